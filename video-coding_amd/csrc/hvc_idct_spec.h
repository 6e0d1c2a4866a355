/* hvc_idct_spec.h -- the arithmetic of the shipped decode kernel (k_decode_packed, k_decode_444) AS DATA.
 *
 * One list of operations describes a 1-D Chen-Wang pass (jpeg/model/src/dct.ml:11-54 rows, :56-98 columns) in
 * the packed-operand form of hvc_kernels.hip.  Two readers:
 *   - hvc_kernels.hip expands HVC_IDCT_PASS into the statements of idct_row_packed / idct_col_packed (the
 *     kernel's arithmetic IS this list: there is no second copy to drift away from it);
 *   - tests/test_guard_bounds.py parses THIS FILE (the #defines and the list) and replays the same operations
 *     on intervals: the proof that no int32 operation wraps and no int16 operand pair is inexact under the
 *     guards below is a proof about the code that ships.
 * Keep every value a plain integer expression of other names in this file: the parser evaluates nothing else.
 */
#ifndef HVC_IDCT_SPEC_H
#define HVC_IDCT_SPEC_H

/* dct.ml:4-9 */
#define HVC_W1 2841
#define HVC_W2 2676
#define HVC_W3 2408
#define HVC_W5 1609
#define HVC_W6 1108
#define HVC_W7 565

/* The four operand pairs of a pass: positions (lo half, hi half) inside the row (or column) they come from.
 * A = (x4, x5), B = (x6, x7), C = (x3, x2), Z = (b0, b4) in dct.ml's names. */
#define HVC_PAIR_A_LO 1
#define HVC_PAIR_A_HI 7
#define HVC_PAIR_B_LO 5
#define HVC_PAIR_B_HI 3
#define HVC_PAIR_C_LO 2
#define HVC_PAIR_C_HI 6
#define HVC_PAIR_Z_LO 0
#define HVC_PAIR_Z_HI 4

/* Pass parameters.  Row pass (dct.ml:11-54): x0 = b0 << 11 + 128, rotations unrounded and unshifted, outputs >> 8.
 * Column pass (:56-98): x0 = b0 << 8 + 8192, rotations (+ 4) >> 3, outputs >> 14 -- that last shift is done by
 * the store stage's saturating pack (v_ashr_pk_u8_i32), together with recon's + 128 (decoder.ml:220), which
 * rides in the addend as 128 << 14: so the column pass leaves its outputs unshifted here. */
#define HVC_ROW_ZSCALE 2048
#define HVC_ROW_ZADD 128
#define HVC_ROW_RADD 0
#define HVC_ROW_RSHIFT 0
#define HVC_ROW_OSHIFT 8
#define HVC_COL_ZSCALE 256
#define HVC_COL_ZADD (8192 + (128 << 14))
#define HVC_COL_RADD 4
#define HVC_COL_RSHIFT 3
#define HVC_COL_OSHIFT 0
#define HVC_COL_PACK_SHIFT 14
/* x2 = (181 * (x4 + x5) + 128) >> 8, x4 = (181 * (x4 - x5) + 128) >> 8   (dct.ml:41-42, 83-84) */
#define HVC_M181_MUL 181
#define HVC_M181_ADD 128
#define HVC_M181_SHIFT 8

/* Guards of the packed kernel (anything outside goes to the int64 kernel):
 *   coefficient energy  E = SUM c^2 <= (HVC_GUARD_D_PACKED / qmax)^2      => every |c * q| <= HVC_GUARD_D_PACKED
 *   row-output energy   SUM sat16(r)^2 < HVC_GUARD_RE                      => every |r| < 32767, the pack was exact
 *   |argument of a 181 * y product| <= HVC_GUARD_Y                         (v_mad_i32_i24 operand) */
#define HVC_GUARD_D_PACKED 32767
#define HVC_GUARD_RE (32767 * 32767)
#define HVC_GUARD_Y ((1 << 23) - 1)

/* The pass.  ROT(d, P, klo, khi): d = (P.lo * klo + P.hi * khi + RADD) >> RSHIFT        (v_dot2_i32_i16)
 *            ZDOT(d, P, slo, shi): d = P.lo * slo * ZSCALE + P.hi * shi * ZSCALE + ZADD   (v_dot2_i32_i16)
 *            ADD / SUB(d, a, b); GUARDY(a, b): both must lie within +-HVC_GUARD_Y;
 *            M181(d, a): d = (M181_MUL * a + M181_ADD) >> M181_SHIFT                      (v_mad_i32_i24)
 *            OUTADD / OUTSUB(i, a, b): output i = (a +- b) >> OSHIFT */
#define HVC_IDCT_PASS(ROT, ZDOT, ADD, SUB, GUARDY, M181, OUTADD, OUTSUB)                              \
    ROT(n4, A, HVC_W1, HVC_W7)   ROT(n5, A, HVC_W7, -HVC_W1)                                          \
    ROT(n6, B, HVC_W5, HVC_W3)   ROT(n7, B, HVC_W3, -HVC_W5)                                          \
    ROT(n3, C, HVC_W2, HVC_W6)   ROT(n2, C, HVC_W6, -HVC_W2)                                          \
    ZDOT(e8, Z, 1, 1)            ZDOT(e0, Z, 1, -1)                                                   \
    ADD(x1, n4, n6)  ADD(x6, n5, n7)  SUB(x4, n4, n6)  SUB(x5, n5, n7)                                \
    ADD(x7, e8, n3)  SUB(x8, e8, n3)  ADD(x3, e0, n2)  SUB(x0, e0, n2)                                \
    ADD(ys, x4, x5)  SUB(yd, x4, x5)  GUARDY(ys, yd)                                                  \
    M181(x2, ys)     M181(y4, yd)                                                                     \
    OUTADD(0, x7, x1) OUTADD(1, x3, x2) OUTADD(2, x0, y4) OUTADD(3, x8, x6)                           \
    OUTSUB(4, x8, x6) OUTSUB(5, x0, y4) OUTSUB(6, x3, x2) OUTSUB(7, x7, x1)

#endif /* HVC_IDCT_SPEC_H */
