#!/usr/bin/env python3
"""Mutated and truncated reference files through the HOST header parser and Huffman reader (no GPU needed): meant
to be run against an AddressSanitizer build of the library's host code,

    make -C video-coding_amd/csrc -f Makefile.asan asan
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
        HVC_JPEG_LIB=video-coding_amd/libhvc_asan.so python tools/fuzz_host.py

Every stream either decodes or is rejected with an hvc error; a memory error aborts the process.
"""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np
import video_coding_amd as hvc
from conftest import golden_bytes
import argparse
_ap = argparse.ArgumentParser()
_ap.add_argument("--seed", type=int, default=5)
_ap.add_argument("--scale", type=int, default=1, help="multiplies the number of cases of every part")
_args = _ap.parse_args()
rng = np.random.Generator(np.random.PCG64(_args.seed))
base = [golden_bytes("mini.jpg"), golden_bytes("Mouse480.jpg")]
ok = err = 0
for it in range(6000 * _args.scale):
    b = bytearray(base[it & 1])
    mode = it % 5
    if mode == 4:   # truncation
        b = b[:int(rng.integers(2, len(b)))]
    else:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    b = bytes(b)
    try:
        info = hvc.hvc.jpeg_read_header(b)
        if info.coef_count > 1 << 24: continue
        hvc.hvc.jpeg_entropy_decode(b, info)
        ok += 1
    except hvc.HvcError:
        err += 1
print("decoded", ok, "rejected", err)
# two files in turn (hvc_jpeg_entropy_decode2): a mutated stream beside an intact one -- every pair gives each file the
# status and the record the single-file entry point gives it
pair_ok = pair_err = 0
whole = [hvc.hvc.jpeg_entropy_decode(b)[1] for b in base]
for it in range(1500 * _args.scale):
    b = bytearray(base[it & 1])
    if it % 4 == 3:
        b = b[:int(rng.integers(2, len(b)))]
    else:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    b = bytes(b)
    try:
        info = hvc.hvc.jpeg_read_header(b)
        if info.coef_count > 1 << 24: continue
    except hvc.HvcError:
        continue
    try:
        want, rec = 0, hvc.hvc.jpeg_entropy_decode(b, info)[1]
    except hvc.HvcError as e:
        want, rec = e.code, None
    other = (it >> 1) & 1
    (sa, _, ra), (sb, _, rb) = hvc.hvc.jpeg_entropy_decode2(b, base[other]) if it % 3 else hvc.hvc.jpeg_entropy_decode2(base[other], b)[::-1]
    assert sa == want and sb == 0, (it, sa, want, sb)
    assert np.array_equal(rb, whole[other]) and (rec is None or np.array_equal(ra, rec)), it
    pair_ok += want == 0
    pair_err += want != 0
print("pairs: mutated file decoded", pair_ok, "rejected", pair_err)
# the host coder with arbitrary int16 records (values without a code must be refused, never written past a buffer)
enc_ok = enc_err = 0
for it in range(300):
    chroma = int(rng.choice([420, 422, 444]))
    w, h = int(rng.integers(1, 20)) * 2, int(rng.integers(1, 20)) * 2
    try:
        info = hvc.hvc.jpeg_encoder_layout(w, h, chroma, int(rng.integers(1, 101)))
    except hvc.HvcError:
        continue
    lim = int(rng.choice([2, 64, 1024, 2048, 32768]))
    rec = rng.integers(-lim, lim, size=info.coef_count).astype(np.int16)
    if it % 3 == 0:
        rec[rng.integers(0, 2, size=rec.size) == 0] = 0
    try:
        jpg = hvc.hvc.jpeg_entropy_encode(info, rec)
        _, back = hvc.hvc.jpeg_entropy_decode(jpg)
        assert np.array_equal(back, rec), "coder / reader round trip"
        enc_ok += 1
    except hvc.HvcError:
        enc_err += 1
print("encoded", enc_ok, "refused", enc_err)
# round 4: the reader with restart intervals honoured (the opt-in extension) on mutated files that carry DRI / RSTn, files with
# an empty plane, files whose tables name DC categories of up to 62 bits -- decoded or refused, never a memory error
sys.path.insert(0, os.path.join(_ROOT, 'tools'))
from jpeg_opt_writer import jpeg_optimised_tables
qt = np.stack([np.arange(1, 65), np.arange(64, 0, -1)]).astype(np.uint16)
seeds = []
for sampling, w, h, ri in (([(2, 2), (1, 1), (1, 1)], 96, 64, 2), ([(1, 1)] * 3, 40, 24, 1), ([(2, 1), (1, 1), (0, 1)], 64, 32, 3),
                           ([(2, 2), (0, 0), (1, 1)], 48, 48, 0)):
    mh, mv = max(s[0] for s in sampling), max(s[1] for s in sampling)
    Wr, Hr = -(-w // (8 * mh)) * 8 * mh, -(-h // (8 * mv)) * 8 * mv
    nblk = sum((Wr * sh // mh // 8) * (Hr * sv // mv // 8) for sh, sv in sampling)
    rec = np.zeros((nblk, 64), dtype=object)
    rec[:, 0] = [int(x) for x in rng.integers(-500, 501, size=nblk)]
    if ri == 0:
        rec[::7, 0] = [int(x) << 40 for x in rng.integers(-100, 101, size=len(rec[::7]))]    # (categories of 40-odd bits: HVC_E_RANGE at best)
    rec[:, 1:4] = rng.integers(-30, 31, size=(nblk, 3))
    seeds.append(jpeg_optimised_tables(w, h, sampling, qt, rec.reshape(-1), table_sets=1, restart_interval=ri))
r_ok = r_err = 0
for it in range(4000 * _args.scale):
    b = bytearray(seeds[it % len(seeds)])
    if it % 5 == 4:
        b = b[:int(rng.integers(2, len(b)))] + b"\xff\xd9"
    else:
        for _ in range(int(rng.integers(0, 5))):
            pos = int(rng.integers(0, len(b)))
            b[pos] = int(rng.choice([int(rng.integers(0, 256)), 0xFF, 0xD0 + int(rng.integers(0, 8)), 0]))
    b = bytes(b)
    for restart in (False, True):
        try:
            info = hvc.hvc.jpeg_read_header(b)
            if info.coef_count > 1 << 24: continue
            hvc.hvc.jpeg_entropy_decode(b, info, restart_markers=restart)
            r_ok += 1
        except hvc.HvcError:
            r_err += 1
print("restart intervals / empty planes / wide categories: decoded", r_ok, "rejected", r_err)
