"""Exhaustive proof that the encoder kernel's float evaluation of
Encoder.quant_and_scale (jpeg/model/src/encoder.ml:98-101) is exact.

    model : f < 0 ? (f - 2t) / (4t) : (f + 2t) / (4t)        (truncating division)
    kernel: v_cvt_rpi_i32_f32(float(f) * r) = floor(f * r + 0.5),   r = fl32((1 + 2^-16) / (4t))

for every table entry t in 1..255 and every f with |f| <= 2^15 (the forward DCT of
8-bit pixels stays below 2^15: tests/test_guard_bounds.py).  The float32 product is
emulated exactly (a float64 holds the exact product of two float32, rounded once to
float32).  v_cvt_rpi's "+ 0.5" is checked under both possible evaluations -- exact,
and rounded to float32 before the floor -- so the proof does not depend on which one
the hardware implements.  CPU only.
"""
import numpy as np


def kernel_quant(f, t, round_sum_to_f32):
    r = np.float32((1.0 + 1.0 / 65536.0) / (4.0 * t))
    x = (f.astype(np.float64) * np.float64(r)).astype(np.float32)       # v_mul_f32 (v_cvt_f32_i32 is exact)
    s = x.astype(np.float64) + 0.5
    if round_sum_to_f32:
        s = s.astype(np.float32).astype(np.float64)
    return np.floor(s).astype(np.int64)


def model_quant(f, t):
    n = np.where(f < 0, f - 2 * t, f + 2 * t)
    return np.sign(n) * (np.abs(n) // (4 * t))  # truncating division


def test_quant_and_scale_exact_for_all_tables_and_values():
    f = np.arange(-(1 << 15), (1 << 15) + 1, dtype=np.int64)
    for t in range(1, 256):
        want = model_quant(f, t)
        assert np.array_equal(kernel_quant(f, t, False), want), t
        assert np.array_equal(kernel_quant(f, t, True), want), t


def test_quantised_range_fits_int16():
    f = np.array([-(1 << 15), 1 << 15], dtype=np.int64)
    assert np.abs(model_quant(f, 1)).max() < 32768
