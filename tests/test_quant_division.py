"""Exhaustive proof that the encoder kernel's float evaluation of
Encoder.quant_and_scale (jpeg/model/src/encoder.ml:98-101) is exact.

    model : f < 0 ? (f - 2t) / (4t) : (f + 2t) / (4t)        (truncating division)
    kernel: low 16 bits of fma(float(f), r, 1.5 * 2^23),   r = fl32((1 + 2^-16) / (4t))          (shipped, HVC_ENCODE_QMAGIC)
            = the integer nearest to the EXACT product f * r (one rounding, at an ulp of 1; ties to even)
    before: v_cvt_rpi_i32_f32(float(f) * r) = floor(f * r + 0.5)                                   (HVC_ENCODE_QMAGIC=0)

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


def kernel_quant_magic(f, t):
    """v_fma_f32(float(f), r, 12582912.0): the product is exact inside the fma (16 x 24 bits: a float64 holds it), the sum is
    rounded once to a float32 whose ulp is 1 -- 12582912 + rint(f * r), 12582912 being even -- and the kernel keeps the low
    16 bits of the bit pattern 0x4B400000 + q (v_perm_b32).  Returns (q, number of exact ties met)."""
    r = np.float32((1.0 + 1.0 / 65536.0) / (4.0 * t))
    p = f.astype(np.float64) * np.float64(r)                      # exact
    y = (np.float64(12582912.0) + np.rint(p)).astype(np.float32)  # representable: an integer below 2^24
    bits = y.view(np.uint32).astype(np.int64)
    q16 = bits & 0xFFFF
    return np.where(q16 >= 0x8000, q16 - 0x10000, q16), int(np.sum(p - np.floor(p) == 0.5))


def model_quant(f, t):
    n = np.where(f < 0, f - 2 * t, f + 2 * t)
    return np.sign(n) * (np.abs(n) // (4 * t))  # truncating division


def test_quant_and_scale_exact_for_all_tables_and_values():
    f = np.arange(-(1 << 15), (1 << 15) + 1, dtype=np.int64)
    for t in range(1, 256):
        want = model_quant(f, t)
        assert np.array_equal(kernel_quant(f, t, False), want), t
        assert np.array_equal(kernel_quant(f, t, True), want), t
        got, ties = kernel_quant_magic(f, t)
        assert ties == 0 and np.array_equal(got, want), t          # (no tie: round-to-even never decides)


def test_quantised_range_fits_int16():
    f = np.array([-(1 << 15), 1 << 15], dtype=np.int64)
    assert np.abs(model_quant(f, 1)).max() < 32768
