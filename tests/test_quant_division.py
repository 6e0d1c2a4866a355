"""Exhaustive proof that the encoder kernel's float evaluation of
Encoder.quant_and_scale (jpeg/model/src/encoder.ml:98-101) is exact.

    model : f < 0 ? (f - 2t) / (4t) : (f + 2t) / (4t)      (truncating division)
    kernel: trunc(copysign(fma(|f|, r, h), f)),  r = fl32(1/(4t)),  h = fl32(0.5 + 0.5/(4t))

for every table entry t in 1..255 and every f with |f| <= 2^15 (the forward DCT
of 8-bit pixels stays below 2^15: tests/test_guard_bounds.py).  fmaf is
emulated exactly: the exact product-sum fits a float64 (<= 46 significant bits)
and is then rounded once to float32.  CPU only.
"""
import numpy as np


def kernel_quant(f, t):
    d = np.float32(4.0) * np.float32(t)
    r = np.float32(1.0) / d
    h = np.float32(0.5) + np.float32(0.5) / d
    x = (np.abs(f).astype(np.float64) * np.float64(r) + np.float64(h)).astype(np.float32)  # == fmaf
    q = np.trunc(x).astype(np.int64)
    return np.where(f < 0, -q, q)


def model_quant(f, t):
    n = np.where(f < 0, f - 2 * t, f + 2 * t)
    return np.sign(n) * (np.abs(n) // (4 * t))  # truncating division


def test_quant_and_scale_exact_for_all_tables_and_values():
    f = np.arange(-(1 << 15), (1 << 15) + 1, dtype=np.int64)
    for t in range(1, 256):
        assert np.array_equal(kernel_quant(f, t), model_quant(f, t)), t


def test_quantised_range_fits_int16():
    f = np.array([-(1 << 15), 1 << 15], dtype=np.int64)
    assert np.abs(model_quant(f, 1)).max() < 32768
