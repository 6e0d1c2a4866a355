"""Interval-arithmetic proofs for the int32 / 24-bit-multiply fast kernels
(video-coding_amd/csrc/hvc_kernels.hip).  CPU only.

Every operation of idct_1d_fast / fdct_1d is replayed on intervals.  The tests
show: under the kernel's range guard (GUARD_D, GUARD_R, GUARD_Y) no int32
operation can wrap and every v_mul_i32_i24 / v_mad_i32_i24 operand lies in
[-2^23, 2^23); and the encoder needs no guard at all because its inputs are
8-bit pixels.  Intervals ignore correlations, so the bounds are conservative.
"""
import re
import os

I32 = (-(1 << 31), (1 << 31) - 1)
I24 = (-(1 << 23), (1 << 23) - 1)

W1, W2, W3, W5, W6, W7 = 2841, 2676, 2408, 1609, 1108, 565


class Iv:
    """closed integer interval that asserts it fits int32 on construction"""

    def __init__(self, lo, hi, what=""):
        assert lo <= hi
        assert I32[0] <= lo and hi <= I32[1], "int32 overflow possible in %s: [%d, %d]" % (what, lo, hi)
        self.lo, self.hi = lo, hi

    def __add__(self, o):
        o = o if isinstance(o, Iv) else Iv(o, o)
        return Iv(self.lo + o.lo, self.hi + o.hi, "add")

    def __sub__(self, o):
        o = o if isinstance(o, Iv) else Iv(o, o)
        return Iv(self.lo - o.hi, self.hi - o.lo, "sub")

    def __neg__(self):
        return Iv(-self.hi, -self.lo)

    def shl(self, k):
        return Iv(self.lo << k, self.hi << k, "shl")

    def asr(self, k):
        return Iv(self.lo >> k, self.hi >> k)

    def clampto(self, lo, hi):
        """guard: the kernel only continues when the value is inside [lo, hi]"""
        return Iv(max(self.lo, lo), min(self.hi, hi))

    def absmax(self):
        return max(abs(self.lo), abs(self.hi))


def mul24(c, x):
    assert I24[0] <= c <= I24[1]
    assert I24[0] <= x.lo and x.hi <= I24[1], "mul24 operand outside 24 bits: [%d, %d]" % (x.lo, x.hi)
    a, b = c * x.lo, c * x.hi
    return Iv(min(a, b), max(a, b), "mul24")


def mad24(c, x, acc):
    acc = acc if isinstance(acc, Iv) else Iv(acc, acc)
    return mul24(c, x) + acc


def kernel_constants():
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-coding_amd", "csrc")
    src = open(os.path.join(csrc, "hvc_kernels.hip")).read()
    hdr = open(os.path.join(csrc, "hvc_kernels.h")).read()
    out = {}
    m = re.search(r"constexpr int GUARD_R = \(1 << (\d+)\) - 1;", src)
    out["GUARD_R"] = (1 << int(m.group(1))) - 1
    assert "constexpr int GUARD_Y = HVC_GUARD_Y;" in src
    out["GUARD_Y"] = idct_spec()[0]["HVC_GUARD_Y"]
    m = re.search(r"#define HVC_GUARD_D \(\(1 << (\d+)\) - 1\)", hdr)
    out["GUARD_D"] = (1 << int(m.group(1))) - 1
    assert "constexpr int GUARD_D = HVC_GUARD_D;" in src
    return out


def idct_1d_fast(b, col, bias, guard_y):
    """mirror of idct_1d_fast<COL, BIAS>; returns the 8 outputs (column pass: unshifted)"""
    R = 4 if col else 0
    x0 = b[0].shl(8) + (8192 + bias) if col else b[0].shl(11) + 128
    x1 = b[4].shl(8) if col else b[4].shl(11)
    x2, x3, x4, x5, x6, x7 = b[6], b[2], b[1], b[7], b[5], b[3]
    n4 = mad24(W7, x5, mad24(W1, x4, R))
    n5 = mad24(-W1, x5, mad24(W7, x4, R))
    n6 = mad24(W3, x7, mad24(W5, x6, R))
    n7 = mad24(-W5, x7, mad24(W3, x6, R))
    n3 = mad24(W6, x2, mad24(W2, x3, R))
    n2 = mad24(-W2, x2, mad24(W6, x3, R))
    if col:
        n4, n5, n6, n7, n3, n2 = (v.asr(3) for v in (n4, n5, n6, n7, n3, n2))
    x8 = x0 + x1
    x0 = x0 - x1
    x1 = n4 + n6
    x4 = n4 - n6
    x6 = n5 + n7
    x5 = n5 - n7
    x7 = x8 + n3
    x8 = x8 - n3
    x3 = x0 + n2
    x0 = x0 - n2
    ys = (x4 + x5).clampto(-guard_y, guard_y)   # g.y2(ys, yd): outside -> wide kernel
    yd = (x4 - x5).clampto(-guard_y, guard_y)
    x2 = mad24(181, ys, 128).asr(8)
    x4 = mad24(181, yd, 128).asr(8)
    S = 0 if col else 8
    outs = [x7 + x1, x3 + x2, x0 + x4, x8 + x6, x8 - x6, x0 - x4, x3 - x2, x7 - x1]
    return [o.asr(S) for o in outs]


def test_decode_fast_path_cannot_overflow_under_its_guard():
    k = kernel_constants()
    d = Iv(-k["GUARD_D"], k["GUARD_D"])
    # dequantisation: int16 coefficient x 8-bit table entry through v_mul_i32_i24
    assert 32768 * 255 <= I24[1], "coefficient * q must fit the 24-bit multiplier"
    rows = idct_1d_fast([d] * 8, col=False, bias=0, guard_y=k["GUARD_Y"])
    r = Iv(-k["GUARD_R"], k["GUARD_R"])          # g.r2(): row outputs outside -> wide kernel
    rows = [x.clampto(r.lo, r.hi) for x in rows]
    cols = idct_1d_fast([r] * 8, col=True, bias=128 << 14, guard_y=k["GUARD_Y"])
    for c in cols:   # consumed by v_ashr_pk_u8_i32 (arithmetic shift of an int32): any int32 is fine
        assert I32[0] <= c.lo and c.hi <= I32[1]


def test_energy_threshold_bounds_every_dequantised_coefficient():
    """hvc_capi.hip: ethr = min((GUARD_D / qmax)^2, 2^31 - 2) and the kernel flags E > ethr
    (E saturates at 2^31 - 1).  So an accepted block has max|c| <= sqrt(E) <= GUARD_D / qmax,
    hence |c * q| <= GUARD_D; when ethr is capped, qmax <= 2 and 32768 * qmax <= GUARD_D anyway."""
    k = kernel_constants()
    import math
    for qmax in list(range(1, 256)) + [256, 1000, 65535]:
        m = k["GUARD_D"] // qmax
        thr = min(m * m, 0x7FFFFFFE)
        if thr == m * m:
            cmax = math.isqrt(thr)            # largest |c| with c^2 <= thr
            assert cmax * qmax <= k["GUARD_D"]
        else:
            assert 32768 * qmax <= k["GUARD_D"]


def test_guard_is_not_vacuous_and_has_headroom_for_real_data():
    """Encoder-producible blocks peak far below the guard: |dequant| <= ~1.1k*q-rounding,
    row outputs <= ~3e4, 181-arguments <= ~7.8e6 < GUARD_Y = 8388607."""
    k = kernel_constants()
    assert k["GUARD_D"] >= 16 * 2048 and k["GUARD_R"] >= 4 * 32768 and k["GUARD_Y"] == (1 << 23) - 1


def fdct_1d(p):
    def c4(f, g):
        return mul24(362, f + g).asr(9)

    def c4m(f, g):
        return mul24(362, f - g).asr(9)

    def c62(f, g):
        return mad24(473, g, mul24(196, f)).asr(9)

    def c71(f, g):
        return mad24(502, g, mul24(100, f)).asr(9)

    def c35(f, g):
        return mad24(284, g, mul24(426, f)).asr(9)

    a0, c3 = p[0] + p[7], p[0] - p[7]
    a1, c2 = p[1] + p[6], p[1] - p[6]
    a2, c1 = p[2] + p[5], p[2] - p[5]
    a3, c0 = p[3] + p[4], p[3] - p[4]
    b0, b1, b2, b3 = a0 + a3, a1 + a2, a1 - a2, a0 - a3
    o = [None] * 8
    o[0], o[4], o[2], o[6] = c4(b0, b1), c4m(b0, b1), c62(b2, b3), c62(b3, -b2)
    b0, b1 = c4m(c2, c1), c4(c2, c1)
    a0, a1, a2, a3 = c0 + b0, c0 - b0, c3 - b1, c3 + b1
    o[1], o[5], o[3], o[7] = c71(a0, a3), c35(a1, a2), c35(a2, -a1), c71(a3, -a0)
    return o


def mulhi24(c, x):
    """v_mul_hi_i32_i24: the high dword of the signed 24 x 24 -> 48-bit product = floor(c x / 2^32); both operands must be
    genuine 24-bit values"""
    assert I24[0] <= c <= I24[1], "mulhi24 constant outside 24 bits"
    assert I24[0] <= x.lo and x.hi <= I24[1], "mulhi24 operand outside 24 bits: [%d, %d]" % (x.lo, x.hi)
    a, b = c * x.lo, c * x.hi
    return Iv(min(a, b) >> 32, max(a, b) >> 32, "mulhi24")


def fdct_1d_shipped(p, level_shift):
    """the butterfly as k_encode issues it since round 5 (fdct_tail with HVC_ENCODE_MULHI, csrc/hvc_kernels.hip): c4 as one
    v_mul_hi_i32_i24 of an operand pre-shifted by 9 with the constant 362 << 14 -- every pre-shifted operand must stay inside
    24 bits.  level_shift: the sums still carry +256 each (column pass), taken out as -1024 << 9 in front of the one c4 that
    sees them."""
    C4S = 362 << 14
    ref = fdct_1d([x - 128 for x in p] if level_shift else p)   # (the model's form: what the outputs are compared with)
    a0, c3 = p[0] + p[7], p[0] - p[7]
    a1, c2 = p[1] + p[6], p[1] - p[6]
    a2, c1 = p[2] + p[5], p[2] - p[5]
    a3, c0 = p[3] + p[4], p[3] - p[4]
    b0s, b1s, b2, b3 = (a0 + a3).shl(9), (a1 + a2).shl(9), a1 - a2, a0 - a3
    o = [None] * 8
    o[0] = mulhi24(C4S, b0s + b1s - (1024 << 9) if level_shift else b0s + b1s)
    o[4] = mulhi24(C4S, b0s - b1s)
    c2s = c2.shl(9)
    e0, e1 = mulhi24(C4S, mad24(-512, c1, c2s)), mulhi24(C4S, mad24(512, c1, c2s))
    for k in (0, 4):   # same interval as the model's (362 x) >> 9 on the same sums
        assert (o[k].lo, o[k].hi) == (ref[k].lo, ref[k].hi), k
    return ref, (e0, e1)


def test_c4_by_mul_hi_operands_fit_24_bits_in_both_passes():
    """K3's c4 = v_mul_hi_i32_i24((x << 9), 362 << 14) is exact while |x << 9| < 2^23: replayed on intervals for the column pass
    (un-shifted pixels 0..255, the -1024 << 9 folded in) and the row pass (whatever the column pass can produce); and pointwise,
    (x << 9) * (362 << 14) >> 32 == (362 x) >> 9 over the whole range a sum can take"""
    assert (362 << 14) <= I24[1]
    cols, _ = fdct_1d_shipped([Iv(0, 255)] * 8, True)
    m1 = max(c.absmax() for c in cols)
    rows, _ = fdct_1d_shipped([Iv(-m1, m1)] * 8, False)
    assert max(c.absmax() for c in rows) <= 1 << 15
    for x in list(range(-(1 << 14) + 1, 1 << 14, 7)) + [-(1 << 14) + 1, (1 << 14) - 1, -1, 0, 1]:
        assert ((x << 9) * (362 << 14)) >> 32 == (362 * x) >> 9, x


def test_encode_path_needs_no_guard():
    p = Iv(-128, 127)
    cols = fdct_1d([p] * 8)
    m1 = max(c.absmax() for c in cols)
    rows = fdct_1d([Iv(-m1, m1)] * 8)
    m2 = max(c.absmax() for c in rows)
    # the quantiser's float path is verified exhaustively for |f| <= 2^15 (test_quant_division.py)
    assert m2 <= 1 << 15, m2
    # quantised values fit int16 trivially
    assert (m2 + 2) // 4 + 1 < 32768


# ---------------------------------------------------------------------------
# k_decode_packed: int16 operand pairs + v_dot2_i32_i16
I16 = (-32768, 32767)


def dot2(pair, k, add=0):
    """v_dot2_i32_i16: both halves must be genuine int16 values; the sum must fit int32."""
    (a, b), (ka, kb) = pair, k
    for x in (a, b):
        assert I16[0] <= x.lo and x.hi <= I16[1], "dot2 operand outside int16: [%d, %d]" % (x.lo, x.hi)
    assert all(I16[0] <= c <= I16[1] for c in (ka, kb))
    pa = sorted((ka * a.lo, ka * a.hi))
    pb = sorted((kb * b.lo, kb * b.hi))
    return Iv(pa[0] + pb[0] + add, pa[1] + pb[1] + add, "dot2")


# The packed kernel's pass is not mirrored here by hand: video-coding_amd/csrc/hvc_idct_spec.h holds it as data
# (constants + one operation list), hvc_kernels.hip expands that list into the kernel's statements, and this
# file parses the same header and replays the same list on intervals.
def idct_spec():
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-coding_amd", "csrc")
    text = open(os.path.join(csrc, "hvc_idct_spec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments out
    text = text.replace("\\\n", " ")                            # line continuations joined
    consts = {}
    ops = None
    for m in re.finditer(r"^#define[ \t]+(\w+)(\([^)]*\))?[ \t]+(.*)$", text, flags=re.M):
        name, params, body = m.group(1), m.group(2), m.group(3).strip()
        if name == "HVC_IDCT_PASS":
            assert params.replace(" ", "") == "(ROT,ZDOT,ADD,SUB,GUARDY,M181,OUTADD,OUTSUB)"
            ops = [(o.group(1), [a.strip() for a in o.group(2).split(",")]) for o in re.finditer(r"(\w+)\(([^()]*)\)", body)]
        elif params is None and body and name != "HVC_IDCT_SPEC_H":
            assert re.fullmatch(r"[\w\s()+\-*<]+", body), (name, body)  # integer expressions of earlier names only
            consts[name] = int(eval(body, {"__builtins__": {}}, dict(consts)))
    assert ops and len(ops) == 29, ops
    return consts, ops


def replay_pass(consts, ops, pairs, col, guard_y):
    """the operation list of hvc_idct_spec.h on intervals; pairs = {"A": (lo, hi), ...} of Iv operands"""
    pre = "HVC_COL_" if col else "HVC_ROW_"
    K = lambda n: consts[pre + n]
    val = lambda t: consts[t] if t in consts else (-consts[t[1:]] if t.startswith("-") and t[1:] in consts else int(t))
    env, out = {}, [None] * 8
    for op, a in ops:
        if op == "ROT":
            env[a[0]] = dot2(pairs[a[1]], (val(a[2]), val(a[3])), K("RADD")).asr(K("RSHIFT"))
        elif op == "ZDOT":
            env[a[0]] = dot2(pairs[a[1]], (val(a[2]) * K("ZSCALE"), val(a[3]) * K("ZSCALE")), K("ZADD"))
        elif op == "ADD":
            env[a[0]] = env[a[1]] + env[a[2]]
        elif op == "SUB":
            env[a[0]] = env[a[1]] - env[a[2]]
        elif op == "GUARDY":   # g.y2(): a block whose value lies outside goes to the int64 kernel
            for n in a:
                env[n] = env[n].clampto(-guard_y, guard_y)
        elif op == "M181":
            env[a[0]] = mad24(consts["HVC_M181_MUL"], env[a[1]], consts["HVC_M181_ADD"]).asr(consts["HVC_M181_SHIFT"])
        elif op == "OUTADD":
            out[int(a[0])] = (env[a[1]] + env[a[2]]).asr(K("OSHIFT"))
        elif op == "OUTSUB":
            out[int(a[0])] = (env[a[1]] - env[a[2]]).asr(K("OSHIFT"))
        else:
            raise AssertionError("unknown operation %s in hvc_idct_spec.h" % op)
    assert all(o is not None for o in out)
    return out


def test_spec_header_is_what_the_kernel_compiles():
    """hvc_kernels.hip must take its packed passes from the list (no hand-written second copy), the host side its
    pair order, and the constants must be the model's (dct.ml:4-9, :11-98)."""
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-coding_amd", "csrc")
    src = open(os.path.join(csrc, "hvc_kernels.hip")).read()
    capi = open(os.path.join(csrc, "hvc_capi.hip")).read()
    row = src[src.index("void idct_row_packed("):src.index("// saturating pack of two row outputs")]
    col = src[src.index("void idct_col_packed("):src.index("// two adjacent pixels of a row")]
    assert "HVC_EXPAND_PASS(ROW)" in row and "HVC_EXPAND_PASS(COL)" in col
    for body in (row, col):   # no arithmetic of its own next to the expansion
        assert "dot2<" not in body and "mad24(" not in body and " >> " not in body
    assert "HVC_IDCT_PASS(HVC_OP_ROT_##PASS" in src and "HVC_PAIR_A_LO, HVC_PAIR_A_HI" in capi
    assert "constexpr int GUARD_RE = HVC_GUARD_RE;" in src and "constexpr int GUARD_Y = HVC_GUARD_Y;" in src
    c, ops = idct_spec()
    assert [c["HVC_W%d" % i] for i in (1, 2, 3, 5, 6, 7)] == [W1, W2, W3, W5, W6, W7]
    assert (c["HVC_ROW_ZSCALE"], c["HVC_ROW_ZADD"], c["HVC_ROW_OSHIFT"]) == (1 << 11, 128, 8)            # dct.ml:14-15, 45-53
    assert (c["HVC_COL_ZSCALE"], c["HVC_COL_RADD"], c["HVC_COL_RSHIFT"]) == (1 << 8, 4, 3)               # dct.ml:59-60, 67-76
    assert c["HVC_COL_ZADD"] == 8192 + (128 << c["HVC_COL_PACK_SHIFT"]) and c["HVC_COL_PACK_SHIFT"] == 14  # :59, 89-97 + recon's 128
    # the list computes the model's pass: exact integers on a block of values (against the literal dct.ml butterfly)
    import random
    rnd = random.Random(5)
    for colpass in (False, True):
        for _ in range(200):
            b = [rnd.randint(-900, 900) for _ in range(8)]
            P = lambda lo, hi: (Iv(b[lo], b[lo]), Iv(b[hi], b[hi]))
            pairs = {k: P(c["HVC_PAIR_%s_LO" % k], c["HVC_PAIR_%s_HI" % k]) for k in "ABCZ"}
            got = [o.lo for o in replay_pass(c, ops, pairs, colpass, c["HVC_GUARD_Y"])]
            want = model_pass(b, colpass)
            if colpass:  # the list leaves the column outputs unshifted, with recon's 128 riding along
                got = [(g >> 14) - 128 for g in got]
            assert got == want, (colpass, b)


def model_pass(b, col):
    """dct.ml:11-54 (row) / :56-98 (column), literally"""
    if col:
        x0, x1 = (b[0] << 8) + 8192, b[4] << 8
    else:
        x0, x1 = (b[0] << 11) + 128, b[4] << 11
    x2, x3, x4, x5, x6, x7 = b[6], b[2], b[1], b[7], b[5], b[3]
    r, s = (4, 3) if col else (0, 0)
    x8 = W7 * (x4 + x5) + r
    x4 = (x8 + (W1 - W7) * x4) >> s
    x5 = (x8 - (W1 + W7) * x5) >> s
    x8 = W3 * (x6 + x7) + r
    x6 = (x8 - (W3 - W5) * x6) >> s
    x7 = (x8 - (W3 + W5) * x7) >> s
    x8 = x0 + x1
    x0 = x0 - x1
    x1 = W6 * (x3 + x2) + r
    x2 = (x1 - (W2 + W6) * x2) >> s
    x3 = (x1 + (W2 - W6) * x3) >> s
    x1 = x4 + x6
    x4 = x4 - x6
    x6 = x5 + x7
    x5 = x5 - x7
    x7 = x8 + x3
    x8 = x8 - x3
    x3 = x0 + x2
    x0 = x0 - x2
    x2 = (181 * (x4 + x5) + 128) >> 8
    x4 = (181 * (x4 - x5) + 128) >> 8
    sh = 14 if col else 8
    return [v >> sh for v in (x7 + x1, x3 + x2, x0 + x4, x8 + x6, x8 - x6, x0 - x4, x3 - x2, x7 - x1)]


def test_packed_kernel_cannot_overflow_under_its_guard():
    """E <= (32767/qmax)^2 => every dequantised coefficient is an exact int16 (v_pk_mul_lo_u16 keeps
    the low 16 bits of c*q, which are the value itself when it fits); the row outputs may be anything
    in int32 (they are saturate-packed), and the row-energy guard renergy < 32767^2 -- computed on
    the saturated values, so one clipped value alone reaches it -- leaves |r| < 32767: exact pairs.
    The operations replayed are the ones hvc_idct_spec.h lists, i.e. the ones the kernel executes."""
    c, ops = idct_spec()
    assert c["HVC_GUARD_D_PACKED"] == 32767 and c["HVC_GUARD_RE"] == 32767 * 32767 and c["HVC_GUARD_Y"] == (1 << 23) - 1
    gy = c["HVC_GUARD_Y"]
    d = Iv(-c["HVC_GUARD_D_PACKED"], c["HVC_GUARD_D_PACKED"])
    rows = replay_pass(c, ops, {k: (d, d) for k in "ABCZ"}, False, gy)
    for o in rows:  # v_cvt_pk_i16_i32 saturates any int32
        assert I32[0] <= o.lo and o.hi <= I32[1]
    # a saturated half is +-32767/-32768 and contributes >= 32767^2 to renergy: flagged
    assert 32767 * 32767 >= c["HVC_GUARD_RE"] and 32768 * 32768 >= c["HVC_GUARD_RE"]
    rmax = math_isqrt(c["HVC_GUARD_RE"] - 1)      # largest |r| an accepted block can hold
    assert rmax == 32766
    r = Iv(-rmax, rmax)
    cols = replay_pass(c, ops, {k: (r, r) for k in "ABCZ"}, True, gy)
    for o in cols:  # consumed by v_ashr_pk_u8_i32
        assert I32[0] <= o.lo and o.hi <= I32[1]


def math_isqrt(n):
    import math
    return math.isqrt(n)


def test_packed_energy_thresholds():
    import math
    for qmax in range(1, 256):
        m = 32767 // qmax
        thr = min(m * m, 0x7FFFFFFE)
        assert math.isqrt(thr) * qmax <= 32767
    # real-data headroom: a full-amplitude block has sum(pixel^2) <= 64 * 128^2, its row outputs
    # carry energy ~ 511x that (22.6^2): about half of the 32767^2 threshold.
    assert 511 * 64 * 128 * 128 < 32767 * 32767


def test_avg4_by_lerp_identity():
    """k_decode_444 computes Planar_444.avg4 (tools/src/planar_444.ml:10-16) on four samples per
    instruction as v_lerp_u8(avg2(a, b), (c + d) >> 1, r) with r = ~(a ^ b) | (c ^ d) (bit 0):
    ((a+b+1)>>1 + (c+d)>>1 + r) >> 1 == (a+b+c+d+2) >> 2 for every a, b, c, d in 0..255.
    Both sides depend on (a, b) and (c, d) only through their sums, so all 511 x 511 sums cover it."""
    import numpy as np
    s1, s2 = np.meshgrid(np.arange(511), np.arange(511), indexing="ij")
    hc = (s1 + 1) >> 1            # avg2(a, b), v_lerp_u8 with rounding bit 1
    hf = s2 >> 1                  # v_lerp_u8 with rounding bit 0
    r = (~s1 | s2) & 1            # parity(a ^ b) = parity(a + b)
    assert np.array_equal((hc + hf + r) >> 1, (s1 + s2 + 2) >> 2)


def test_q16_exchange_step_is_covered_by_the_packed_proof():
    """k_decode_q16 (one block per quarter wavefront) evaluates exactly the expressions of
    idct_1d_packed, split over an even and an odd lane; the one operation it adds is the two's-complement
    negation of the odd lane's four values before the cross-lane add (theirs + (-mine) = E - O).
    Under the same guards those values are far from INT_MIN, so the negation is exact."""
    gy = kernel_constants()["GUARD_Y"]

    def odd_lane_values(A, B, col):
        rnd, rs = (4, 3) if col else (0, 0)
        n4, n5 = dot2(A, (W1, W7), rnd).asr(rs), dot2(A, (W7, -W1), rnd).asr(rs)
        n6, n7 = dot2(B, (W5, W3), rnd).asr(rs), dot2(B, (W3, -W5), rnd).asr(rs)
        x1, x6, x4, x5 = n4 + n6, n5 + n7, n4 - n6, n5 - n7
        ys = (x4 + x5).clampto(-gy, gy)
        yd = (x4 - x5).clampto(-gy, gy)
        return [x1, x6, mad24(181, ys, 128).asr(8), mad24(181, yd, 128).asr(8)]

    d, r = Iv(-32767, 32767), Iv(-32766, 32766)
    for vals in (odd_lane_values((d, d), (d, d), False), odd_lane_values((r, r), (r, r), True)):
        for v in vals:
            assert v.lo > I32[0] and -v.hi >= I32[0] and -v.lo <= I32[1]
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-coding_amd", "csrc",
                            "hvc_kernels.hip")).read()
    # the kernel uses the packed kernel's thresholds, not its own
    q16 = src[src.index("void k_decode_q16("):src.index("// K1 wide: the model")]
    assert "P.ethr_packed[br.qtab]" in q16 and "GUARD_RE" in q16 and "GUARD_Y" in q16


def test_xcd_work_is_a_permutation_of_the_grid():
    """csrc/hvc_kernels.hip xcd_work: workgroup id = y * tiles + x -> ((k / R) * 8 + id % 8) * R + k % R with k = id / 8, inside
    the whole groups of 8 R workgroups; frame = umulhi(linear, ceil(2^32 / tiles)).  Replayed here with the kernel's own
    integer steps: every (frame, tile) of the grid is taken exactly once, for the benches' grids and for awkward ones, and
    the reciprocal is exact wherever xcd_map_for lets the mapping on (tiles * tiles * frames < 2^32).  (That the kernels
    really cover every tile is what the GPU parity tests at full size see: a tile nobody took would stay zero.)"""
    import numpy as np

    def mapped(per, n, sh):
        magic = ((1 << 32) + per - 1) // per
        group, total = 8 << sh, per * n
        full = total - total % group
        ids = np.arange(total, dtype=np.uint64)
        k = ids >> np.uint64(3)
        lin = ((((k >> np.uint64(sh)) << np.uint64(3)) + (ids & np.uint64(7))) << np.uint64(sh)) + (k & np.uint64((1 << sh) - 1))
        lin = np.where(ids < full, lin, ids)
        frame = (lin * np.uint64(magic)) >> np.uint64(32)
        assert np.array_equal(frame, lin // np.uint64(per))            # the reciprocal is exact on this grid
        return lin

    for per, n in ((192, 1024), (1521, 128), (192, 7), (5, 3), (761, 256), (192, 9), (3, 1000), (64, 64)):
        assert per * per * n < 1 << 32
        for sh in (0, 3, 4, 5, 6, 9):
            lin = mapped(per, n, sh)
            assert np.array_equal(np.sort(lin), np.arange(per * n, dtype=np.uint64)), (per, n, sh)
    # what it is for: inside a whole group, the workgroups with id % 8 == x (one XCD's) take runs of R consecutive positions
    lin = mapped(192, 1024, 5)
    for xcd in range(8):
        mine = lin[xcd::8][:96]
        assert np.array_equal(mine[:32], np.arange(32) + xcd * 32) and np.array_equal(mine[32:64], np.arange(32) + 256 + xcd * 32)


def test_wide_kernel_without_the_63_bit_reading():
    """k_decode_wide_all computes in plain int64 (asr63<S, false>: one shift, no 63-bit reading).  That equals the model's
    63-bit arithmetic while every value of both passes stays below 2^62 in magnitude; replayed here on magnitudes (sums of
    absolute values: conservative) for the kernel's inputs -- an int16 coefficient (or an int16 DC from the compact array) times
    a 16-bit table entry.  The text of idct8_wide is parsed for the operations, so the replay follows the source."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video-coding_amd", "csrc", "hvc_kernels.hip")).read()
    body = src[src.index("void idct8_wide("):src.index("// decoder.ml:142-149 `coefs.(i) * qnt_tab.(i)`")]
    # the statements the replay below restates, in the source's order (a change of the pass breaks this list first)
    for stmt in ("u64 x8 = w7 * (x4 + x5) + R;", "x4 = asr63<RS, WRAP63>(x8 + (w1 - w7) * x4);", "x5 = asr63<RS, WRAP63>(x8 - (w1 + w7) * x5);",
                 "x8 = w3 * (x6 + x7) + R;", "x6 = asr63<RS, WRAP63>(x8 - (w3 - w5) * x6);", "x7 = asr63<RS, WRAP63>(x8 - (w3 + w5) * x7);",
                 "x1 = w6 * (x3 + x2) + R;", "x2 = asr63<RS, WRAP63>(x1 - (w2 + w6) * x2);", "x3 = asr63<RS, WRAP63>(x1 + (w2 - w6) * x3);",
                 "x2 = asr63<8, WRAP63>(181u * ys + 128u);", "x4 = asr63<8, WRAP63>(181u * yd + 128u);",
                 "o[0] = (int64_t)asr63<S, WRAP63>(x7 + x1);", "o[7] = (int64_t)asr63<S, WRAP63>(x7 - x1);"):
        assert stmt in body, stmt
    assert "decode_block_wide<false>(w, P.qt + br.qtab * 64, P.dc_plane != nullptr, dc, out);" in src      # the all-blocks kernel
    assert src.count("decode_block_wide<true>(") == 2                                                          # the two list kernels

    LIMIT = 1 << 62
    seen = [0]

    def chk(v):
        assert v < LIMIT, v
        seen[0] = max(seen[0], v)
        return v

    def one_pass(b, col):   # magnitudes in, magnitudes out
        s_in, r0, r, rs, outshift = (256, 8192, 4, 3, 14) if col else (2048, 128, 0, 0, 8)
        x0, x1 = chk(b[0] * s_in + r0), chk(b[4] * s_in)
        x2, x3, x4, x5, x6, x7 = b[6], b[2], b[1], b[7], b[5], b[3]
        t = chk(W7 * (x4 + x5) + r)
        x4, x5 = chk(t + (W1 - W7) * x4) >> rs, chk(t + (W1 + W7) * x5) >> rs
        t = chk(W3 * (x6 + x7) + r)
        x6, x7 = chk(t + (W3 - W5) * x6) >> rs, chk(t + (W3 + W5) * x7) >> rs
        x8, x0 = chk(x0 + x1), chk(x0 + x1)
        t = chk(W6 * (x3 + x2) + r)
        x2, x3 = chk(t + (W2 + W6) * x2) >> rs, chk(t + (W2 - W6) * x3) >> rs
        x1, x4n, x6n, x5n = chk(x4 + x6), chk(x4 + x6), chk(x5 + x7), chk(x5 + x7)
        x7n, x8n, x3n, x0n = chk(x8 + x3), chk(x8 + x3), chk(x0 + x2), chk(x0 + x2)
        y = chk(181 * (x4n + x5n) + 128) >> 8
        out = max(chk(x7n + x1), chk(x3n + y), chk(x0n + y), chk(x8n + x6n)) >> outshift
        return [out + 1] * 8   # (+ 1: a floor of a negative value can be one larger in magnitude)

    rows = one_pass([32768 * 65535] * 8, col=False)
    one_pass(rows, col=True)
    assert seen[0] < 1 << 57   # 32 times below the wrap-around: the plain shift is the model's asr
