"""Shared test helpers (CPU side).  The oracle is the checker."""
import numpy as np

from oracle import orc


def coef_planes_from_jpeg(data):
    """Run the oracle's Sequenced decoder over a JPEG and collect, per component,
    the quantised coefficients in the C-ABI layout ([bh][bw][64] int16, zig-zag,
    DC absolute), the quantiser table and the decoded (padded) plane."""
    d = orc.Decoder(data)
    dims = [d.info(i) for i in range(d.ncomp)]
    coefs = [np.zeros((dims[i]["decoded_height"] // 8, dims[i]["decoded_width"] // 8, 64), dtype=np.int16)
             for i in range(d.ncomp)]
    while True:
        ci = d.next_block()
        if ci is None:
            break
        inf = d.info(ci)
        c = d.array(ci, "coefs").copy()
        c[0] = inf["dc_pred"]
        coefs[ci][inf["y"] // 8, inf["x"] // 8] = c
    return [dict(coefs=coefs[i], qtab=d.array(i, "quant_table").astype(np.uint16), plane=d.plane(i), info=dims[i])
            for i in range(d.ncomp)], d


def pcg32(seed, n):
    """Small deterministic generator (numpy PCG64 seeded) -> uint32 array."""
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)


def synth_pixels(seed, h, w):
    """Synthetic pixel plane (h, w multiples of 8): per 8x8 block either a smooth
    ramp with low noise or uniform random bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bh, bw = h // 8, w // 8
    kind = rng.integers(0, 2, size=(bh, bw))
    yy, xx = np.mgrid[0:8, 0:8]
    gx = rng.integers(-12, 13, size=(bh, bw))
    gy = rng.integers(-12, 13, size=(bh, bw))
    base = rng.integers(0, 256, size=(bh, bw))
    ramp = (base[:, :, None, None] + gx[:, :, None, None] * xx + gy[:, :, None, None] * yy
            + rng.integers(-3, 4, size=(bh, bw, 8, 8)))
    noise = rng.integers(0, 256, size=(bh, bw, 8, 8))
    blk = np.where(kind[:, :, None, None] == 0, ramp, noise)
    blk = np.clip(blk, 0, 255).astype(np.uint8)
    return blk.transpose(0, 2, 1, 3).reshape(h, w)


def synth_coefs(seed, bh, bw, qtab):
    """Valid (encoder-producible) coefficient plane [bh][bw][64] via the oracle's forward path."""
    pix = synth_pixels(seed, bh * 8, bw * 8)
    return orc.fdct_quant(pix, qtab, bw, bh).reshape(bh, bw, 64), pix


# Baseline JPEG files with per-file optimised Huffman tables: tools/jpeg_opt_writer.py (pure Python, no oracle)
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "tools"))
from jpeg_opt_writer import jpeg_optimised_tables  # noqa: E402,F401


CHECKSUM_MUL = np.uint64(0x9E3779B97F4A7C15)


def checksum_records(records):
    """include/hvc_jpeg.h hvc_checksum_records on the CPU: records uint8 [n][bytes] -> uint64 [n]."""
    records = np.ascontiguousarray(records, dtype=np.uint8)
    records = records.reshape(records.shape[0], -1)
    with np.errstate(over="ignore"):
        w = (np.arange(records.shape[1], dtype=np.uint64) * np.uint64(2) + np.uint64(1)) * CHECKSUM_MUL
        return ((records.astype(np.uint64) + np.uint64(1)) * w[None, :]).sum(axis=1, dtype=np.uint64)


def config1_frame():
    """BASELINE.json config 1's 128x128 4:2:0 frame: the reference's mini64x64.420 tiled 2x2"""
    from conftest import golden_bytes
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    return tuple(np.ascontiguousarray(np.tile(p, (2, 2))) for p in (y, u, v))


def every_symbol_record(info):
    """A coefficient record (int16, info.coef_count) in which EVERY symbol of the default Huffman tables occurs in every
    component: the 160 (run, size) AC symbols with size 1..10 (one block each: one coefficient of magnitude 2^(size-1) ...
    2^size - 1 behind `run` zeros), EOB (every block that does not end at position 63), ZRL (runs of 16 ... 62 zeros), and the
    twelve DC categories (differences of +-(2^(k-1)) ... between consecutive blocks of a component).  Needs >= 176 blocks per
    component (a 4:4:4 frame of 128 x 88)."""
    rec = np.zeros(info.coef_count, dtype=np.int16)
    for i in range(info.n_comp):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h
        assert n >= 176, n
        blk = rec[L.coef_offset:L.coef_offset + n * 64].reshape(n, 64)
        b = 0
        for run in range(16):
            for size in range(1, 11):
                mag = (1 << (size - 1)) + (run * 7 + size) % (1 << (size - 1))      # any value of that size
                blk[b, 1 + run] = mag if (run + size) & 1 else -mag
                b += 1
        for k, gap in enumerate((16, 17, 31, 32, 33, 47, 48, 61, 62)):                # ZRL chains; the last one ends at 63
            blk[b, 1 + gap] = 3 + k
            b += 1
        # DC: category k difference between consecutive blocks, signs alternating so that the values stay small
        dc, vals = 0, []
        for j in range(n):
            k = j % 12
            d = 0 if k == 0 else (1 << (k - 1)) + (j % (1 << (k - 1)))
            dc += d if dc <= 0 else -d
            vals.append(dc)
        blk[:, 0] = np.array(vals, dtype=np.int16)
    return rec
