import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import video_coding_amd as hvc
from video_coding_amd.synth import synth_frame_pixels
W, H, n = 1920, 1080, 512
planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
ctx = hvc.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
src = torch.from_numpy(np.stack([synth_frame_pixels(90 + 8 * f, planes) for f in range(4)])).cuda()
d_distinct = torch.zeros((4, tcfs), dtype=torch.int16, device="cuda")
ctx.encode_frames(src, tpfs, qtabs, hvc.hvc.components(tspecs), 4, d_distinct, tcfs)
ref = None
for rep in range(2):
    for align, ofs_align in ((1, 1), (65536, 1), (65536, 65536), (65536, 2 << 20), (1, 65536)):
        specs, cfs, _ = hvc.hvc.frame_layout(planes, align=align)
        d_coefs = hvc.hvc.spread_records(d_distinct, tspecs, specs, cfs, "coef_offset").repeat(n // 4, 1).contiguous()
        ofs = (3 * W * H + ofs_align - 1) // ofs_align * ofs_align
        d_out = torch.zeros((n, ofs), dtype=torch.uint8, device="cuda")
        comps = hvc.hvc.components(specs)
        for _ in range(20):
            ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out, ofs)
        torch.cuda.synchronize()
        ctx.timer_begin()
        for _ in range(40):
            ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out, ofs)
        ms = ctx.timer_end() / 40
        algo = n * ((240 * 135 + 2 * 120 * 68) * 128 + 3 * W * H)
        out = d_out[:4, :3 * W * H].clone()
        if ref is None:
            ref = out
        print("fused 4:4:4  coef planes align %7d, output frame stride align %7d: %.4f ms  %.2f %%  same frames: %s"
              % (align, ofs_align, ms, 100 * algo / (ms * 1e-3) / 8e12, bool(torch.equal(out, ref))), flush=True)
ctx.close()
