#!/usr/bin/env python3
"""A SUSTAINED run of the headline step (bench.py's config 2: 1080p 4:2:0, 1024 frames = 9.6 GB per launch): at least
--seconds of back-to-back launches (>= --min-launches), every launch bracketed by its own pair of HIP events on the
kernel's stream, so that clock or thermal drift over the run is visible: first / last decile of the per-launch times,
min / median / max, and the slope.  The short timed region of bench.py (50 launches, 80 ms) cannot show that.

    python tools/bench_sustained.py [--seconds 3] [--min-launches 1200] [--frames 1024]

One JSON line.  (The events bracket the whole hvc_decode_frames call on the stream: k_decode_packed plus the
~4 us fix-up kernel behind it.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--min-launches", type=int, default=1200)
    ap.add_argument("--frames", type=int, default=1024)
    args = ap.parse_args()
    import torch
    import video_coding_amd as hvc
    wl = bench.WORKLOADS[2]
    planes = wl["planes"]
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_distinct, qtabs = bench.make_distinct_frames(ctx, hvc, planes, 8, wl["seed"], 0)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    d_coefs = d_distinct.repeat((args.frames + 7) // 8, 1)[:args.frames].contiguous()
    d_pix = torch.zeros((args.frames, pfs), dtype=torch.uint8, device="cuda")
    for _ in range(10):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, args.frames, d_pix, pfs)
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    while len(events) < args.min_launches or time.perf_counter() - t0 < args.seconds:
        for _ in range(100):  # enqueue in batches: the host stays ahead of the GPU without an unbounded queue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.decode_frames(d_coefs, cfs, qtabs, comps, args.frames, d_pix, pfs)
            e1.record()
            events.append((e0, e1))
        events[-1][1].synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = np.array([a.elapsed_time(b) for a, b in events])
    n = len(ms)
    dec = max(1, n // 10)
    algo = args.frames * sum(bw * bh for bw, bh, _ in planes) * 192
    gbps = lambda m: algo / (m * 1e-3) / 1e9
    slope = float(np.polyfit(np.arange(n), ms, 1)[0]) * n  # ms of drift over the whole run (linear fit)
    sums = [int(x) for x in ctx.checksum_records(d_pix, pfs, 8)]
    want = bench.expected_checksums(2, 8, 0)
    print(json.dumps({
        "what": "sustained run of bench.py's config-2 step", "launches": n, "wall_s": round(wall, 3), "frames_per_launch": args.frames,
        "call_ms": {"first_decile_mean": round(float(ms[:dec].mean()), 4), "last_decile_mean": round(float(ms[-dec:].mean()), 4),
                    "min": round(float(ms.min()), 4), "median": round(float(np.median(ms)), 4), "max": round(float(ms.max()), 4),
                    "mean": round(float(ms.mean()), 4), "linear_drift_over_run_ms": round(slope, 4)},
        "algorithmic_GBps": {"first_decile": round(gbps(ms[:dec].mean()), 1), "last_decile": round(gbps(ms[-dec:].mean()), 1),
                             "mean": round(gbps(ms.mean()), 1)},
        "frac_of_8TBps_mean": round(gbps(ms.mean()) / 8000, 4),
        "gpu_busy_fraction": round(float(ms.sum()) * 1e-3 / wall, 3),
        "Mpixel_s": round(n * args.frames * 1920 * 1080 / wall / 1e6, 1),
        "verified": want is not None and sums == want}))
    ctx.close()


if __name__ == "__main__":
    main()
