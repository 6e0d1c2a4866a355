#!/usr/bin/env python3
"""Randomised sweep over the file-level decode pipelines: hvc_jpeg_decode_batch_gpu (GPU Huffman reader, two reader
streams, downloader thread) against hvc_jpeg_decode_batch (host reader) -- both product paths, the second pinned to
the model restatement by the test suite -- over batch sizes, chunk sizes, thread counts, host / device output and
the fused 4:4:4 form.  One summary line; exit code 1 on a mismatch.

    python tools/stress_pipeline.py [--cases 60] [--seed 3]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import video_coding_amd as hvc  # noqa: E402
from video_coding_amd.synth import synth_pixels  # noqa: E402
from jpeg_opt_writer import jpeg_optimised_tables  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=3)
    args = ap.parse_args()
    import torch
    rng = np.random.Generator(np.random.PCG64(args.seed))
    ctx = hvc.Context(0)
    bad = odd = own_batches = odd_samplings = empty_planes = restart_batches = 0
    for case in range(args.cases):
        w = int(rng.integers(1, 30)) * 16
        h = int(rng.integers(1, 20)) * 16
        q = int(rng.choice([10, 50, 75, 95]))
        n_distinct = int(rng.integers(1, 5))
        files = []
        for d in range(n_distinct):
            s = int(rng.integers(0, 1 << 30))
            files.append(ctx.jpeg_encode(synth_pixels(s, h, w), synth_pixels(s + 1, h // 2, w // 2), synth_pixels(s + 2, h // 2, w // 2),
                                         w, h, 420, q))
        sampling = None
        if rng.integers(0, 4) == 0:     # a quarter of the batches: sampling factors the encoder never writes (any factors 1..4,
            ncomp = int(rng.integers(1, 5))  # one to four components), random sparse coefficient records
            sampling = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(ncomp)]
            if ncomp > 1 and rng.integers(0, 3) == 0:   # ... a component without blocks (a factor of zero, never the first one's):
                k = int(rng.integers(1, ncomp))          # the model's empty plane -- both pipelines walk around it
                sampling[k] = (0, sampling[k][1]) if rng.integers(0, 2) else (sampling[k][0], 0)
                empty_planes += 1
            restart = int(rng.integers(1, 9)) if rng.integers(0, 2) == 0 else 0   # restart intervals, honoured on request (the extension)
            mh, mv = max(a for a, _ in sampling), max(b for _, b in sampling)
            Wr, Hr = -(-w // (8 * mh)) * 8 * mh, -(-h // (8 * mv)) * 8 * mv
            nblk = sum((Wr * a // mh // 8) * (Hr * b // mv // 8) for a, b in sampling)
            qt = np.stack([hvc.hvc.quant_table(0, q), hvc.hvc.quant_table(1, q)])
            files = []
            for d in range(n_distinct):
                blocks = np.zeros((nblk, 64), dtype=np.int16)
                blocks[:, 0] = rng.integers(-200, 201, size=nblk)
                dense = rng.random((nblk, 63)) < rng.choice([0.02, 0.1, 0.4])
                blocks[:, 1:][dense] = rng.integers(-50, 51, size=int(dense.sum()))
                files.append(jpeg_optimised_tables(w, h, sampling, qt, blocks.reshape(-1), min(len(sampling), int(rng.integers(1, 4))) if not any(0 in f for f in sampling) else 1,
                                                   restart_interval=restart))
            odd_samplings += 1
            restart_batches += restart > 0
            ctx.set_restart_markers(restart > 0 and bool(rng.integers(0, 2)))   # (off: the model's reading of the marked files)
        own = int(rng.integers(0, 3)) if sampling is None else 0   # a third of the batches: some or all files re-written with their own optimised Huffman
        if own == 1:                    # tables (1, 2 or 3 table sets): the GPU pipeline's per-frame-table mode
            qt = np.stack([hvc.hvc.quant_table(0, q), hvc.hvc.quant_table(1, q)])
            for d in range(n_distinct):
                if n_distinct == 1 or rng.integers(0, 4) != 0:
                    files[d] = jpeg_optimised_tables(w, h, 420, qt, hvc.hvc.jpeg_entropy_decode(files[d])[1], int(rng.integers(1, 4)),
                                                     ac_shape="many_prefixes" if rng.integers(0, 3) == 0 else None)
            own_batches += 1
        n = int(rng.integers(1, 41))
        batch = [files[int(rng.integers(0, n_distinct))] for _ in range(n)]
        info = hvc.hvc.jpeg_read_header(batch[0])
        if rng.integers(0, 3) == 0:   # a few streams that end early: their chunks go to the host reader, the others do not
            for _ in range(int(rng.integers(1, 3))):
                f = int(rng.integers(1, n)) if n > 1 else 0
                cut = info.ecs_offset + int(rng.integers(1, max(2, len(batch[f]) - info.ecs_offset - 2)))
                batch[f] = batch[f][:cut] + b"\xff\xd9"
            odd += 1
        yuv444 = bool(rng.integers(0, 2)) and sampling is None
        fs = 3 * w * h if yuv444 else info.pixel_bytes
        chunk = int(rng.integers(1, 10))
        threads = int(rng.integers(1, 9))
        host_out = bool(rng.integers(0, 2))
        if host_out:
            a = np.zeros(n * fs, np.uint8)
            b = np.zeros(n * fs, np.uint8)
        else:
            a = torch.zeros(n * fs, dtype=torch.uint8, device="cuda")
            b = torch.zeros(n * fs, dtype=torch.uint8, device="cuda")
        try:
            ctx.jpeg_decode_batch(batch, b, fs, threads=threads, frames_per_chunk=chunk, yuv444=yuv444, gpu_entropy=False)
            err = None
        except hvc.HvcError as e:
            err = e.code
        try:
            ctx.jpeg_decode_batch(batch, a, fs, threads=threads, frames_per_chunk=chunk, yuv444=yuv444, gpu_entropy=True)
            gerr = None
        except hvc.HvcError as e:
            gerr = e.code
        ctx.set_restart_markers(False)
        if err != gerr:
            bad += 1
            print("ERROR CODES DIFFER", case, err, gerr, file=sys.stderr)
            continue
        if err is not None:
            continue
        same = np.array_equal(a, b) if host_out else bool(torch.equal(a, b))
        if not same:
            bad += 1
            print("MISMATCH", case, (w, h, q, n, chunk, threads, host_out, yuv444), file=sys.stderr)
    print({"cases": args.cases, "batches_with_truncated_files": odd, "batches_with_per_file_tables": own_batches,
           "batches_with_unusual_samplings": odd_samplings, "with_an_empty_plane": empty_planes, "with_restart_markers": restart_batches,
           "mismatches": bad})
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
