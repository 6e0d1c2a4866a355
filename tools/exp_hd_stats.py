#!/usr/bin/env python3
"""Experiment: how the synchronisation rounds of the GPU Huffman reader go on the benchmark's files -- entries of
k_hd_sync's work lists per round, walks and inner rounds of the k_hd_round launches behind them.  Needs a library
built with -DHVC_HD_STATS (make -C video-coding_amd/csrc OUT=../../build/variants/libhvc_hdstats.so EXTRA=-DHVC_HD_STATS),
which prints the counters to stderr after every hvc_jpeg_entropy_decode_gpu call.

    HVC_JPEG_LIB=build/variants/libhvc_hdstats.so python tools/exp_hd_stats.py [--own-tables]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import video_coding_amd as hvc  # noqa: E402
from video_coding_amd.synth import synth_pixels  # noqa: E402


def main():
    W, H = 1920, 1080
    ctx = hvc.Context(0)
    for f in range(4):
        y = synth_pixels(10 + f, 1088, 1920)[:H]
        u = synth_pixels(20 + f, 544, 960)[:H // 2]
        v = synth_pixels(30 + f, 544, 960)[:H // 2]
        j = ctx.jpeg_encode(y, u, v, W, H, 420, 75)
        if "--own-tables" in sys.argv:
            from jpeg_opt_writer import jpeg_optimised_tables
            qt = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
            j = jpeg_optimised_tables(W, H, 420, qt, hvc.hvc.jpeg_entropy_decode(j)[1])
        print("file", f, len(j), "bytes, 16 copies:", file=sys.stderr, flush=True)
        _, _, used = ctx.jpeg_entropy_decode_gpu([j] * 16, device=True)
        print("  gpu reader used:", used, file=sys.stderr, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
