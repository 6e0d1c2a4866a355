#!/usr/bin/env python3
"""Does the placement of the pixel buffer relative to the coefficient buffer matter to K1?  The bench's workload (1024 x 1080p
4:2:0 per launch) with the output shifted by a few offsets; prints kernel ms / fraction of 8 TB/s per offset, alternating.
Measurement tool only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
import video_coding_amd as hvc  # noqa: E402


def main():
    wl = bench.WORKLOADS[2]
    planes = wl["planes"]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    frames = 1024
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_distinct, qtabs = bench.make_distinct_frames(ctx, hvc, planes, 8, wl["seed"], 0)
    d_coefs = d_distinct.repeat(frames // 8, 1).contiguous()
    slack = 1 << 22
    raw = torch.zeros(frames * pfs + slack, dtype=torch.uint8, device="cuda")
    blocks = sum(bw * bh for bw, bh, _ in planes)
    algo = frames * blocks * 192
    ctx.set_profiling(True)
    offsets = [0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1 << 20, (1 << 20) + 65536 + 4096 + 256, 2 * (1 << 20)]
    for rnd in range(3):
        for off in offsets:
            d_pix = raw[off:off + frames * pfs].view(frames, pfs)
            for _ in range(5):
                ctx.decode_frames(d_coefs, cfs, qtabs, comps, frames, d_pix, pfs)
            torch.cuda.synchronize()
            for _ in range(30):
                ctx.decode_frames(d_coefs, cfs, qtabs, comps, frames, d_pix, pfs)
            torch.cuda.synchronize()
            ms = float(np.mean(ctx.kernel_ms_history(30)))
            print("round %d  pixel base + %8d B   coef ptr %% 2 MiB = %7d, pixel ptr %% 2 MiB = %7d   %.4f ms  %.2f %%" % (
                rnd, off, d_coefs.data_ptr() % (1 << 21), d_pix.data_ptr() % (1 << 21), ms, algo / (ms * 1e-3) / 8e12 * 100), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
