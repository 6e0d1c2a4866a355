"""The host Huffman reader by itself (no GPU, no pipeline): hvc_jpeg_entropy_decode / hvc_jpeg_entropy_decode2 on T threads,
each decoding its own copy of the bench's 1080p files into its own records -- what config 3's host stage can deliver at
most, to hold against what the pipeline (tools/bench_configs.py --config 3) gets out of the same threads.

  python tools/bench_host_reader.py [--threads 1,8,16,32] [--seconds 2] [--own-tables]
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import video_coding_amd as pkg  # noqa: E402

H = pkg.hvc


def bench_files(n, own_tables):
    """the files tools/bench_configs.py --config 3 decodes: synthetic 1080p 4:2:0 frames through the library's own encoder
    (quality 75; that one needs the GPU), optionally re-written with Huffman tables optimised per file"""
    from video_coding_amd.synth import synth_pixels
    W, Hh = 1920, 1080
    ctx = pkg.Context(0)
    jpegs = []
    for f in range(n):
        y = synth_pixels(10 + f, 1088, 1920)[:Hh]
        u = synth_pixels(20 + f, 544, 960)[:Hh // 2]
        v = synth_pixels(30 + f, 544, 960)[:Hh // 2]
        jpegs.append(ctx.jpeg_encode(y, u, v, W, Hh, 420, 75))
    ctx.close()
    if own_tables:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from jpeg_opt_writer import jpeg_optimised_tables
        qt = np.stack([H.quant_table(0, 75), H.quant_table(1, 75)])
        jpegs = [jpeg_optimised_tables(W, Hh, 420, qt, H.jpeg_entropy_decode(j)[1]) for j in jpegs]
    return jpegs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,8,16,32")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--own-tables", action="store_true")
    a = ap.parse_args()
    files = bench_files(a.distinct, a.own_tables)
    L = H.lib()
    infos = [H.jpeg_read_header(f) for f in files]
    px = infos[0].width * infos[0].height
    out = []
    for T in [int(t) for t in a.threads.split(",")]:
        for pairs in (1, 0):
            counts = [0] * T
            stop = time.perf_counter() + a.seconds
            errs = []

            def work(t):
                fa, fb = files[(2 * t) % len(files)], files[(2 * t + 1) % len(files)]
                ia, ib = infos[(2 * t) % len(files)], infos[(2 * t + 1) % len(files)]
                ca = np.zeros(ia.coef_count, np.int16)
                cb = np.zeros(ib.coef_count, np.int16)
                sa, sb = C.c_int(), C.c_int()
                n = 0
                while time.perf_counter() < stop:
                    if pairs:
                        r = L.hvc_jpeg_entropy_decode2(fa, len(fa), C.byref(ia), ca.ctypes.data, C.byref(sa),
                                                       fb, len(fb), C.byref(ib), cb.ctypes.data, C.byref(sb))
                        if r or sa.value or sb.value:
                            errs.append((r, sa.value, sb.value))
                            return
                    else:
                        for f, i, c in ((fa, ia, ca), (fb, ib, cb)):
                            r = L.hvc_jpeg_entropy_decode(f, len(f), C.byref(i), c.ctypes.data)
                            if r:
                                errs.append(r)
                                return
                    n += 2
                counts[t] = n

            t0 = time.perf_counter()
            th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            dt = time.perf_counter() - t0
            assert not errs, errs
            rate = sum(counts) * px / dt / 1e6
            line = {"threads": T, "two_files_in_turn": bool(pairs), "Mpixel_s": round(rate, 1), "Mpixel_s_per_thread": round(rate / T, 1),
                    "file_bytes": len(files[0]), "own_tables": a.own_tables}
            print(json.dumps(line), flush=True)
            out.append(line)


if __name__ == "__main__":
    main()
