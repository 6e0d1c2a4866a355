#!/usr/bin/env python3
"""The GPU Huffman reader alone on one chunk of the file-level pipeline's size: hvc_jpeg_entropy_decode_gpu over 256
1080p files (the benchmark's 4 distinct files, segments uploaded by the call), one stream, nothing else on the GPU --
so a kernel trace of this command gives per-kernel durations that do not depend on what the other reader stream is doing.

    python tools/bench_reader_chunk.py [--files 256] [--reps 5] [--own-tables]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import video_coding_amd as hvc  # noqa: E402
from video_coding_amd.synth import synth_pixels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--own-tables", action="store_true")
    ap.add_argument("--restart-interval", type=int, default=0, help="files with DRI / RSTn every so many MCUs (and their own tables), honoured")
    args = ap.parse_args()
    W, H = 1920, 1080
    ctx = hvc.Context(0)
    jpegs = []
    for f in range(4):
        y = synth_pixels(10 + f, 1088, 1920)[:H]
        u = synth_pixels(20 + f, 544, 960)[:H // 2]
        v = synth_pixels(30 + f, 544, 960)[:H // 2]
        j = ctx.jpeg_encode(y, u, v, W, H, 420, 75)
        if args.own_tables or args.restart_interval:
            from jpeg_opt_writer import jpeg_optimised_tables
            qt = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
            j = jpeg_optimised_tables(W, H, 420, qt, hvc.hvc.jpeg_entropy_decode(j)[1], restart_interval=args.restart_interval)
        jpegs.append(j)
    batch = [jpegs[i % 4] for i in range(args.files)]
    want = [hvc.hvc.jpeg_entropy_decode(j, restart_markers=args.restart_interval > 0)[1] for j in jpegs]
    ctx.set_restart_markers(args.restart_interval > 0)
    times = []
    for rep in range(args.reps + 1):
        t0 = time.perf_counter()
        _, recs, used = ctx.jpeg_entropy_decode_gpu(batch, device=True)
        times.append((time.perf_counter() - t0) * 1e3)
        assert used == 1, "the GPU reader declined the batch"
    ok = all(np.array_equal(recs[i], want[i % 4]) for i in range(args.files))
    print(json.dumps({"files": args.files, "own_tables": args.own_tables, "restart_interval": args.restart_interval, "call_ms_min": round(min(times[1:]), 3), "call_ms": [round(t, 2) for t in times[1:]],
                      "records_equal_host_reader": bool(ok)}))
    ctx.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
