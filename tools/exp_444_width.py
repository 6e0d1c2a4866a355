#!/usr/bin/env python3
"""Experiment: the fused 4:4:4 kernel's two halves (HVC_444_ONLY=luma|chroma, set by the caller) at different frame
widths -- does the chroma half run closer to the machine when one 64-block tile spans the whole output row?
    HVC_444_ONLY=chroma python tools/exp_444_width.py 1920 1080 512"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    W, H, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    r16 = lambda x: (x + 15) // 16 * 16
    planes = [(r16(W) // 8, r16(H) // 8, 0), (r16(W) // 16, r16(H) // 16, 1), (r16(W) // 16, r16(H) // 16, 1)]
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.from_numpy(np.stack([synth_frame_pixels(90 + 8 * f, planes) for f in range(4)])).cuda()
    d4 = torch.zeros((4, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, pfs, qtabs, comps, 4, d4, cfs)
    d_coefs = d4.repeat((n + 3) // 4, 1)[:n].contiguous()
    d_out = torch.zeros((n, 3 * W * H), dtype=torch.uint8, device="cuda")
    for _ in range(10):
        ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(20):
        ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out)
    ms = ctx.timer_end() / 20
    only = os.environ.get("HVC_444_ONLY", "all")
    cbw, cbh = (W // 2 + 7) // 8, (H // 2 + 7) // 8
    tiles_x = 1 if cbw <= 64 else (cbw - 1 + 62) // 63
    chroma_blocks = 2 * cbh * sum(min(64, cbw - 63 * t) for t in range(tiles_x))
    luma_blocks = ((W + 7) // 8) * ((H + 7) // 8)
    b = {"luma": luma_blocks * 128 + W * H, "chroma": chroma_blocks * 128 + 2 * W * H}
    b["all"] = b["luma"] + b["chroma"]
    print(json.dumps({"W": W, "H": H, "frames": n, "only": only, "ms": round(ms, 4),
                      "GBps_of_the_half": round(n * b.get(only, b["all"]) / (ms * 1e-3) / 1e9, 1),
                      "frac": round(n * b.get(only, b["all"]) / (ms * 1e-3) / 8e12, 4)}))
    ctx.close()


if __name__ == "__main__":
    main()
