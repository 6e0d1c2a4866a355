#!/usr/bin/env python3
"""One file at a time: where does the GPU Huffman reader start to pay?  For 1080p 4:2:0 files of several
qualities (sizes), times (a) the host reader alone, (b) hvc_jpeg_entropy_decode_gpu for that single file
(records stay on the device) and (c) the whole hvc_jpeg_decode call.  One JSON line per file size.
The files come from the library's own encoder (hvc_jpeg_encode); nothing here touches oracle/.

    python tools/bench_single.py [--reps 20]
"""
import argparse
import ctypes as C
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
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import torch
    ctx = hvc.Context(0)
    w, h = 1920, 1080
    y = synth_pixels(11, 1088, 1920)[:h]
    u = synth_pixels(12, 544, 960)[:540]
    v = synth_pixels(13, 544, 960)[:540]
    lib = hvc.hvc.lib()
    for q in (3, 8, 15, 25, 40, 60, 75, 90):
        jpg = ctx.jpeg_encode(y, u, v, w, h, 420, q)
        info = hvc.hvc.jpeg_read_header(jpg)
        # (a) host reader
        t = []
        rec = np.zeros(info.coef_count, dtype=np.int16)  # (one record, reused: no page faults inside the timed call)
        for _ in range(args.reps):
            t0 = time.perf_counter()
            r = lib.hvc_jpeg_entropy_decode(jpg, len(jpg), C.byref(info), rec.ctypes.data)
            t.append(time.perf_counter() - t0)
            assert r == 0
        host_ms = 1e3 * min(t)
        # (b) GPU reader, single file, device output
        out = torch.empty((1, info.coef_count), dtype=torch.int16, device="cuda")
        ptrs = (C.c_void_p * 1)(C.cast(C.c_char_p(jpg), C.c_void_p))
        sizes = (C.c_size_t * 1)(len(jpg))
        used = C.c_int(-1)
        t = []
        for _ in range(args.reps + 3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = lib.hvc_jpeg_entropy_decode_gpu(ctx._h, ptrs, sizes, 1, out.data_ptr(), info.coef_count, 1, C.byref(info),
                                                C.byref(used))
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
            assert r == 0
        gpu_ms = 1e3 * min(t[3:])
        # (c) the whole call
        t = []
        for _ in range(args.reps + 3):
            t0 = time.perf_counter()
            ctx.jpeg_decode(jpg)
            t.append(time.perf_counter() - t0)
        call_ms = 1e3 * min(t[3:])
        print(json.dumps({"quality": q, "file_kB": round(len(jpg) / 1024, 1), "host_reader_ms": round(host_ms, 3),
                          "gpu_reader_ms": round(gpu_ms, 3), "gpu_reader_used": used.value,
                          "hvc_jpeg_decode_ms": round(call_ms, 3)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
