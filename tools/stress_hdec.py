#!/usr/bin/env python3
"""Differential stress of the GPU Huffman reader against the host reader (both are the product; the host
reader is pinned to the model restatement by the CPU suite): random geometries, qualities and contents,
batches of files with different lengths, then random byte mutations of the streams -- same records or the same
error code.  Files come from the library's own encoder.  Prints one summary line; exit code 1 on a mismatch.

    python tools/stress_hdec.py [--cases 300] [--mutations 600] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import video_coding_amd as hvc  # noqa: E402
from jpeg_opt_writer import jpeg_optimised_tables  # noqa: E402


def content(rng, h, w, kind):
    if kind == 0:    # noise
        return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == 1:    # flat with a few rectangles
        p = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
        for _ in range(3):
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            p[y0:y0 + int(rng.integers(1, h + 1)), x0:x0 + int(rng.integers(1, w + 1))] = int(rng.integers(0, 256))
        return p
    if kind == 2:    # smooth gradient + light noise
        yy, xx = np.mgrid[0:h, 0:w]
        g = (yy * int(rng.integers(0, 4)) + xx * int(rng.integers(0, 4)) + int(rng.integers(0, 64))) % 256
        return np.clip(g + rng.integers(-4, 5, size=(h, w)), 0, 255).astype(np.uint8)
    blk = rng.integers(0, 2, size=((h + 7) // 8, (w + 7) // 8))   # blockwise mix of noise and flat
    m = np.kron(blk, np.ones((8, 8), dtype=np.int64))[:h, :w]
    return np.where(m == 1, rng.integers(0, 256, size=(h, w)), 128).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--mutations", type=int, default=600)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.Generator(np.random.PCG64(args.seed))
    ctx = hvc.Context(0)
    used_gpu = fell_back = bad = refused = with_restart = 0
    keep = []
    for case in range(args.cases):
        chroma = int(rng.choice([420, 422, 444]))
        w = int(rng.integers(1, 40)) * (2 if chroma != 444 else 1) + int(rng.integers(0, 2)) * 600
        h = int(rng.integers(1, 40)) * (2 if chroma == 420 else 1) + int(rng.integers(0, 2)) * 300
        q = int(rng.choice([1, 5, 20, 50, 75, 90, 100]))
        cw = w if chroma == 444 else w // 2
        ch = h // 2 if chroma == 420 else h
        files = []
        try:
            for _ in range(int(rng.integers(1, 5))):
                kind = int(rng.integers(0, 4))
                files.append(ctx.jpeg_encode(content(rng, h, w, kind), content(rng, ch, cw, kind),
                                             content(rng, ch, cw, kind), w, h, chroma, q))
        except hvc.HvcError:   # a geometry the model's encoder raises on (hvc_jpeg_encoder_check)
            refused += 1
            continue
        if case % 3 == 2:   # a third of the cases: files with their own optimised Huffman tables, 1 to 3 table sets each
            qt = np.stack([hvc.hvc.quant_table(0, q), hvc.hvc.quant_table(1, q)])
            files = [jpeg_optimised_tables(w, h, chroma, qt, hvc.hvc.jpeg_entropy_decode(j)[1], int(rng.integers(1, 4)),
                                           ac_shape="many_prefixes" if rng.integers(0, 3) == 0 else None)  # (a third: tables past the sub-table limit)
                     if rng.integers(0, 5) else j for j in files]
        rst = case % 4 == 1   # a quarter of the cases: the files carry restart intervals (one DRI for the batch) and both readers
        if rst:               # honour them (the opt-in extension): every interval is a frame of its own to the GPU reader
            qt = np.stack([hvc.hvc.quant_table(0, q), hvc.hvc.quant_table(1, q)])
            ri = int(rng.integers(1, 2 * max(1, w // 16) + 2))
            files = [jpeg_optimised_tables(w, h, chroma, qt, hvc.hvc.jpeg_entropy_decode(j)[1], int(rng.integers(1, 4)), restart_interval=ri)
                     for j in files]
            ctx.set_restart_markers(True)
            with_restart += 1
        _, got, used = ctx.jpeg_entropy_decode_gpu(files, device=bool(case & 1))
        ctx.set_restart_markers(False)
        used_gpu += used == 1
        fell_back += used != 1
        for f, j in enumerate(files):
            _, want = hvc.hvc.jpeg_entropy_decode(j, restart_markers=rst)
            if not np.array_equal(got[f], want):
                bad += 1
                print("MISMATCH case", case, "file", f, (w, h, chroma, q), file=sys.stderr)
        if len(keep) < 40:
            keep.append((files[0], rst))
    agree = errs = 0
    for it in range(args.mutations):
        b, rst = keep[it % len(keep)]
        b = bytearray(b)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if rst and it % 3 == 0:   # (markers lost, doubled, moved: the host reader's, with the same result)
            marks = [i for i in range(len(b) - 1) if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7]
            if marks:
                m = marks[int(rng.integers(0, len(marks)))]
                kind = int(rng.integers(0, 3))
                b = b[:m] + b[m + 2:] if kind == 0 else b[:m] + b[m:m + 2] + b[m:] if kind == 1 else b[:max(2, m - 5)] + b[m:]
        b = bytes(b)
        try:
            info = hvc.hvc.jpeg_read_header(b)
            if info.coef_count > 1 << 23:
                continue
            _, want = hvc.hvc.jpeg_entropy_decode(b, info, restart_markers=rst)
            err = None
        except hvc.HvcError as e:
            want, err = None, e.code
        ctx.set_restart_markers(rst)
        try:
            _, got, _ = ctx.jpeg_entropy_decode_gpu([b])
            gerr = None
        except hvc.HvcError as e:
            got, gerr = None, e.code
        ctx.set_restart_markers(False)
        if err != gerr or (err is None and not np.array_equal(got[0], want)):
            bad += 1
            print("MISMATCH mutation", it, err, gerr, file=sys.stderr)
        agree += err is None
        errs += err is not None
    print({"cases": args.cases, "encoder_refused_geometry": refused, "with_restart_intervals": with_restart, "gpu_reader_used": used_gpu, "host_fallback": fell_back, "mutations_decoded": agree,
           "mutations_rejected": errs, "mismatches": bad})
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
