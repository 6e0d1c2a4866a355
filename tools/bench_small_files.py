"""hvc_jpeg_decode / hvc_jpeg_decode_yuv444 on the reference's own small test files (jpeg/test_data: mini.jpg 64x64... the
call shape of Decoder.decode_a_frame at the size the reference's tests use it): ms per call, best and median of N.
    python tools/bench_small_files.py [--reps 200]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import video_coding_amd as hvc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    ctx = hvc.Context(0)
    for fn in ("mini.jpg", "Mouse480.jpg"):
        data = open(os.path.join(ROOT, "tests", "golden", fn), "rb").read()
        info = hvc.hvc.jpeg_read_header(data)
        for name, call in (("hvc_jpeg_decode", lambda: ctx.jpeg_decode(data)),
                           ("hvc_jpeg_entropy_decode (host reader alone)", lambda: hvc.hvc.jpeg_entropy_decode(data, info))):
            t = []
            for _ in range(a.reps + 5):
                t0 = time.perf_counter()
                call()
                t.append(time.perf_counter() - t0)
            t = sorted(t[5:])
            print(json.dumps({"file": fn, "bytes": len(data), "size": [info.width, info.height], "call": name,
                              "best_ms": round(t[0] * 1e3, 4), "median_ms": round(t[len(t) // 2] * 1e3, 4)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
