#!/usr/bin/env python3
"""A caller looping over SMALL batches: 64-file calls of hvc_jpeg_decode_batch / _gpu in a loop.
Round 2 created and joined `threads` std::threads (+ a downloader) in every call; the workers now live in the context
(csrc/hvc_pool.h).  Prints per configuration the first call (which starts the threads and allocates the rings), the
median and minimum of the later ones, and how many threads the context ever started -- `threads`, not calls x threads.

    python tools/bench_small_batches.py [--files 64] [--calls 40] [--threads 16]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_coding_amd as hvc  # noqa: E402
from video_coding_amd.synth import synth_pixels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=64)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    import torch
    w, h = 1920, 1080
    ctx = hvc.Context(0)
    distinct = []
    for f in range(4):
        y = synth_pixels(300 + f, 1088, 1920)[:h]
        u = synth_pixels(310 + f, 544, 960)[:540]
        v = synth_pixels(320 + f, 544, 960)[:540]
        distinct.append(ctx.jpeg_encode(y, u, v, w, h, 420, 75))
    files = [distinct[i % 4] for i in range(args.files)]
    fs = hvc.hvc.jpeg_read_header(files[0]).pixel_bytes
    out = torch.zeros((args.files, fs), dtype=torch.uint8, device="cuda")
    for gpu in (False, True):
        c = hvc.Context(0)
        t = []
        for _ in range(args.calls):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c.jpeg_decode_batch(files, out, fs, threads=args.threads, frames_per_chunk=16, gpu_entropy=gpu)
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        alive, ever = c.host_threads()
        later = np.array(t[3:]) * 1e3
        print(json.dumps({"what": "%d-file calls of hvc_jpeg_decode_batch%s in a loop" % (args.files, "_gpu" if gpu else ""),
                          "threads": args.threads, "calls": args.calls, "first_call_ms": round(t[0] * 1e3, 2),
                          "later_calls_ms_median": round(float(np.median(later)), 3), "later_calls_ms_min": round(float(later.min()), 3),
                          "Gpixel_s_median": round(args.files * w * h / (float(np.median(later)) * 1e-3) / 1e9, 2),
                          "host_threads_alive": alive, "host_threads_ever_started": ever,
                          "threads_round2_would_have_started": args.calls * (args.threads + (0 if not gpu else 0))}), flush=True)
        c.close()
    ctx.close()


if __name__ == "__main__":
    main()
