# K1 on layouts whose planes / frames start on 4 KiB boundaries (caller's choice: hvc_component.plane_offset, the frame strides)
import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import video_coding_amd as hvc
from video_coding_amd.synth import synth_frame_pixels

def run(name, planes, frames, align):
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    comps = hvc.hvc.components(specs)
    src = torch.from_numpy(np.stack([synth_frame_pixels(40 + 8 * f, planes) for f in range(4)])).cuda()
    d_distinct = torch.zeros((4, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, pfs, qtabs, comps, 4, d_distinct, cfs)
    # aligned layout: every plane's pixel offset and coefficient offset, and both frame strides, rounded up to `align` bytes
    up = lambda x, a: (x + a - 1) // a * a
    aspecs, co, po = [], 0, 0
    for s in specs:
        co, po = up(co * 2, align) // 2, up(po, align)
        aspecs.append(dict(s, coef_offset=co, plane_offset=po))
        co += s["blocks_w"] * s["blocks_h"] * 64
        po += s["blocks_w"] * s["blocks_h"] * 64
    acfs, apfs = up(co * 2, align) // 2, up(po, align)
    d_coefs = torch.zeros((frames, acfs), dtype=torch.int16, device="cuda")
    for s, a in zip(specs, aspecs):
        n = s["blocks_w"] * s["blocks_h"] * 64
        d_coefs[:, a["coef_offset"]:a["coef_offset"] + n] = d_distinct[:, s["coef_offset"]:s["coef_offset"] + n].repeat((frames + 3) // 4, 1)[:frames]
    d_pix = torch.zeros((frames, apfs), dtype=torch.uint8, device="cuda")
    acomps = hvc.hvc.components(aspecs)
    ctx.set_profiling(True)
    for _ in range(15):
        ctx.decode_frames(d_coefs, acfs, qtabs, acomps, frames, d_pix, apfs)
    torch.cuda.synchronize()
    for _ in range(30):
        ctx.decode_frames(d_coefs, acfs, qtabs, acomps, frames, d_pix, apfs)
    torch.cuda.synchronize()
    k_ms = float(np.mean(ctx.kernel_ms_history(30)))
    blocks = sum(bw * bh for bw, bh, _ in planes)
    # tight reference of the output: same pixels
    ref = torch.zeros((4, pfs), dtype=torch.uint8, device="cuda")
    ctx.decode_frames(d_distinct, cfs, qtabs, comps, 4, ref, pfs)
    torch.cuda.synchronize()
    same = all(torch.equal(d_pix[:4, a["plane_offset"]:a["plane_offset"] + s["blocks_w"] * s["blocks_h"] * 64],
                           ref[:, s["plane_offset"]:s["plane_offset"] + s["blocks_w"] * s["blocks_h"] * 64]) for s, a in zip(specs, aspecs))
    print("%-40s align %6d  %.4f ms  %.2f %% of 8 TB/s  same pixels: %s" % (name, align, k_ms, 100 * frames * blocks * 192 / (k_ms * 1e-3) / 8e12, same), flush=True)
    ctx.close()


def run_enc(name, planes, frames, align):
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.from_numpy(np.stack([synth_frame_pixels(60 + 8 * f, planes) for f in range(4)])).cuda()
    up = lambda x, a: (x + a - 1) // a * a
    aspecs, co, po = [], 0, 0
    for s in specs:
        co, po = up(co * 2, align) // 2, up(po, align)
        aspecs.append(dict(s, coef_offset=co, plane_offset=po))
        co += s["blocks_w"] * s["blocks_h"] * 64
        po += s["blocks_w"] * s["blocks_h"] * 64
    acfs, apfs = up(co * 2, align) // 2, up(po, align)
    d_pix = torch.zeros((frames, apfs), dtype=torch.uint8, device="cuda")
    for s, a in zip(specs, aspecs):
        n = s["blocks_w"] * s["blocks_h"] * 64
        d_pix[:, a["plane_offset"]:a["plane_offset"] + n] = src[:, s["plane_offset"]:s["plane_offset"] + n].repeat((frames + 3) // 4, 1)[:frames]
    d_coefs = torch.zeros((frames, acfs), dtype=torch.int16, device="cuda")
    acomps = hvc.hvc.components(aspecs)
    ctx.set_profiling(True)
    for _ in range(15):
        ctx.encode_frames(d_pix, apfs, qtabs, acomps, frames, d_coefs, acfs)
    torch.cuda.synchronize()
    for _ in range(30):
        ctx.encode_frames(d_pix, apfs, qtabs, acomps, frames, d_coefs, acfs)
    torch.cuda.synchronize()
    k_ms = float(np.mean(ctx.kernel_ms_history(30)))
    blocks = sum(bw * bh for bw, bh, _ in planes)
    print("K3 %-37s align %7d  %.4f ms  %.2f %% of 8 TB/s" % (name, align, k_ms, 100 * frames * blocks * 192 / (k_ms * 1e-3) / 8e12), flush=True)
    ctx.close()

for rep in range(2):
    for align in (1, 65536, 262144, 2097152):
        run("1080p 4:2:0 (config 2)", [(240,136,0),(120,68,1),(120,68,1)], 1024, align)
    for align in (1, 65536, 2097152):
        run("4K 4:4:4 (config 4 launch)", [(480,270,0),(480,270,1),(480,270,1)], 128, align)
    for align in (1, 4096, 65536, 2097152):
        run_enc("4K 4:2:0 (config 5)", [(480,270,0),(240,135,1),(240,135,1)], 256, align)
