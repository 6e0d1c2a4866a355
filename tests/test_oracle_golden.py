"""Pins the CPU oracle (oracle/hvc_oracle.c) to the reference's own golden
vectors.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from oracle import orc


def s12(v):  # low 12 bits -> signed
    v = np.asarray(v, dtype=np.int64)
    return np.where(v >= 2048, v - 4096, v)


def s8(v):
    v = np.asarray(v, dtype=np.int64)
    return np.where(v >= 128, v - 256, v)


def test_data_files_md5():
    """md5 of the reference's jpeg/test_data files."""
    want = {"mini.jpg": "f5d3abe93281f346fbf619d2e701e81d", "mini64x64.420": "06f5376b39f0ddd38024d165528d0d33",
            "mini64x64.422": "4d7e084143297086d1082415e5ac155f", "mini64x64.444": "960c90b2d51c13b48af331903db1091e",
            "Mouse480.jpg": "54025171493c18e20a44e6f5a7401352"}
    for fn, md5 in want.items():
        assert hashlib.md5(golden_bytes(fn)).hexdigest() == md5, fn


def test_g1_chen_forward_and_inverse():
    g = golden_json("g1_chen_dct.json")
    fd = orc.fdct_8x8(g["input"])
    # test_chen_dct.ml:53: x > 0 ? (x+2)/4 : (x-2)/4 with truncating division
    fd4 = np.array([(x + 2) // 4 if x > 0 else -((-(x - 2)) // 4) for x in fd.tolist()])
    assert fd4.tolist() == g["fdct_div4_rounded"]
    assert orc.idct_8x8(g["fdct_div4_rounded"]).tolist() == g["idct_of_fdct"]


def test_g5_quant_scale_and_g6_zigzag():
    g = golden_json("g5_quant_tables.json")
    for q, want in g["luma_scaled"].items():
        assert orc.quant_scale(orc.quant_luma(), int(q)).tolist() == want, q
    izz = orc.zigzag_inverse()
    fzz = orc.zigzag_forward()
    assert izz.tolist() == g["izz"]
    assert [int(izz[fzz[i]]) for i in range(64)] == g["fzz"] == list(range(64))
    # out-of-range qualities clip to 1..100 (quant_tables.ml:142)
    assert orc.quant_scale(orc.quant_luma(), 0).tolist() == g["luma_scaled"]["1"]
    assert orc.quant_scale(orc.quant_luma(), 1000).tolist() == g["luma_scaled"]["100"]


def test_g2_mouse480_first_six_blocks_sequenced():
    """For_testing.Sequenced.decode on Mouse480.jpg: coefs / dequant / idct / recon
    of blocks 0..5 equal the model's printed Component.Summary."""
    g = golden_json("g2_mouse480_blocks.json")
    d = orc.Decoder(golden_bytes("Mouse480.jpg"))
    for blk in g["blocks"]:
        ci = d.next_block()
        inf = d.info(ci)
        assert (inf["x"], inf["y"], inf["dc_pred"], inf["identifier"]) == (
            blk["x"], blk["y"], blk["dc_pred_after"], blk["identifier"])
        assert d.array(ci, "coefs").tolist() == s12(blk["coefs_lo12"]).tolist()
        assert d.array(ci, "dequant").tolist() == s12(blk["dequant_lo12"]).tolist()
        assert d.array(ci, "idct").tolist() == s8(blk["idct_lo8"]).tolist()
        assert d.array(ci, "recon").tolist() == blk["recon"]


def test_mouse480_header_and_entropy_segment():
    g = golden_json("mouse480_header.json")
    d = orc.Decoder(golden_bytes("Mouse480.jpg"))
    assert (d.width, d.height, d.ncomp) == (g["width"], g["height"], 3)
    for i, (ident, h, v, tq) in enumerate(g["components_id_h_v_tq"]):
        inf = d.info(i)
        assert (inf["identifier"], inf["hscale"], inf["vscale"], inf["tq"]) == (ident, h, v, tq)
        want = [t for t in g["quant_tables"] if t["table_identifier"] == tq][0]["elements"]
        assert d.array(i, "quant_table").tolist() == want


def test_g3_mini_jpg_is_byte_exact_encoder_output():
    """mini.jpg == Encoder.encode_420 ~quality:75 of mini64x64.420 (byte for byte)."""
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    jpg = orc.encode_yuv(y, u, v, 64, 64, 420, 75)
    assert jpg == golden_bytes("mini.jpg")


def test_g8_header_bytes():
    g = golden_json("g8_header_c420_480x320_q20.json")
    assert orc.write_headers(g["width"], g["height"], g["chroma"], g["quality"]).hex() == g["hex"]


def _enc_dec_psnr(raw, w, h, chroma, quality):
    y, u, v = orc.split_yuv(raw, w, h, chroma)
    jpg = orc.encode_yuv(y, u, v, w, h, chroma, quality)
    planes = orc.decode_a_frame(jpg)
    return [orc.ocaml_float_to_string(orc.psnr(a, b)) for a, b in zip((y, u, v), planes)], planes


@pytest.mark.parametrize("idx", range(5))
def test_g4_psnr_pins(idx):
    c = golden_json("g4_psnr_pins.json")["cases"][idx]
    got, _ = _enc_dec_psnr(golden_bytes(c["file"]), c["width"], c["height"], c["chroma"], c["quality"])
    assert got == c["psnr"]


def test_g4_nonstandard_size_52x44():
    """test-nonstandard-sizes.t: oyuv convert 64x64 -> 52x44 (420 -> 444, crop, -> 420),
    encode q95, decode, PSNR vs the converted source."""
    c = golden_json("g4_psnr_pins.json")["nonstandard"]
    y, u, v = orc.split_yuv(golden_bytes(c["file"]), 64, 64, 420)
    u4, v4 = orc.supersample_hv2(u), orc.supersample_hv2(v)
    w, h = c["width"], c["height"]
    yc, uc, vc = (orc.crop_plane(p, w, h) for p in (y, u4, v4))
    u2, v2 = orc.subsample_hv2(uc, w // 2, h // 2), orc.subsample_hv2(vc, w // 2, h // 2)
    raw = yc.tobytes() + u2.tobytes() + v2.tobytes()
    got, planes = _enc_dec_psnr(raw, w, h, 420, c["quality"])
    assert got == c["psnr"]
    assert [p.shape for p in planes] == [(44, 52), (22, 26), (22, 26)]


def test_g7_upsample_kats():
    g = golden_json("g7_upsample.json")["cases"]
    A = lambda rows: np.array(rows, dtype=np.uint8)
    f444, f420, back = g["444<->420"]
    assert orc.subsample_hv2(A(f444[4:8]), 2, 2).tolist() == f420[4:6]
    assert orc.subsample_hv2(A(f444[8:12]), 2, 2).tolist() == f420[6:8]
    assert orc.supersample_hv2(A(f420[4:6])).tolist() == back[4:8]
    assert orc.supersample_hv2(A(f420[6:8])).tolist() == back[8:12]
    f444, f422, back = g["444<->422"]
    assert orc.subsample_h2(A(f444[4:8]), 2, 4).tolist() == f422[4:8]
    assert orc.subsample_h2(A(f444[8:12]), 2, 4).tolist() == f422[8:12]
    assert orc.supersample_h2(A(f422[4:8])).tolist() == back[4:8]
    assert orc.supersample_h2(A(f422[8:12])).tolist() == back[8:12]


def test_g8_codewords_and_rle():
    g = golden_json("g8_codewords.json")
    L = orc.lib()
    for i, lo, hi, slo, shi in g["size_ranges_i_lo_hi_sizelo_sizehi"]:
        assert (L.orc_enc_size(lo), L.orc_enc_size(hi)) == (slo, shi)
    for value, size, emag, dmag in g["value_size_emag_dmag"]:
        assert L.orc_enc_size(value) == size
        assert L.orc_enc_magnitude(size, value) == emag
        if size:
            assert L.orc_mag(size, emag) == dmag
    import ctypes as C
    for case in golden_json("g8_rle.json")["cases"]:
        q = np.zeros(64, dtype=np.int64)
        for k, v in case["set"]:
            q[k] = v
        runs = (C.c_int * 65)()
        vals = (C.c_int64 * 65)()
        n = L.orc_enc_rle(q.ctypes.data_as(orc.i64p), C.c_int64(0), runs, vals)
        assert [[runs[i], vals[i]] for i in range(n)] == case["rle"], case["name"]


def test_full_frame_decode_matches_sequenced_and_batch_form():
    """decode = iterate decode_seq; and the batch block-stage entry point
    (the cpu_baseline function) reproduces the planes from absolute-DC coefs."""
    for fn in ("mini.jpg", "Mouse480.jpg"):
        d = orc.Decoder(golden_bytes(fn))
        comps = [dict(coefs=[], info=None) for _ in range(d.ncomp)]
        dims = [d.info(i) for i in range(d.ncomp)]
        coef_planes = [np.zeros((dims[i]["decoded_height"] // 8, dims[i]["decoded_width"] // 8, 64), dtype=np.int16)
                       for i in range(d.ncomp)]
        while True:
            ci = d.next_block()
            if ci is None:
                break
            inf = d.info(ci)
            c = d.array(ci, "coefs").copy()
            c[0] = inf["dc_pred"]  # absolute DC
            coef_planes[ci][inf["y"] // 8, inf["x"] // 8] = c
        d2 = orc.Decoder(golden_bytes(fn))
        d2.decode()
        for i in range(d.ncomp):
            assert np.array_equal(d.plane(i), d2.plane(i))
            bh, bw = coef_planes[i].shape[:2]
            out = orc.dequant_idct_recon(coef_planes[i], d.array(i, "quant_table"), bw, bh)
            assert np.array_equal(out.reshape(bh * 8, bw * 8), d.plane(i)), (fn, i)


def test_fdct_quant_batch_matches_encoder_coefs():
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.444"), 64, 64, 444)
    _, coefs = orc.encode_yuv(y, u, v, 64, 64, 444, 75, want_coefs=True)
    ql, qc = orc.quant_scale(orc.quant_luma(), 75), orc.quant_scale(orc.quant_chroma(), 75)
    for plane, q, c in ((y, ql, coefs[0]), (u, qc, coefs[1]), (v, qc, coefs[2])):
        got = orc.fdct_quant(plane, q, 8, 8).reshape(8, 8, 64)
        assert np.array_equal(got, c)


# ---------------------------------------------------------------------------------------------------------------------
# OCaml's 63-bit ints.  No golden vector of the reference reaches them (nothing an encoder writes does); what is held here
# is the restatement's ARITHMETIC: it keeps sums and products modulo 2^64 and reads them as 63-bit numbers at every `asr`
# and compare (oracle/hvc_oracle.c ocaml_int) -- against the plain reading of the language: every operation on `int` is
# taken modulo 2^63 (two's complement), written out below with Python's unbounded integers and a wrap after EVERY
# operation, statement for statement from dct.ml:11-107 and decoder.ml:142-149, 213-224.

def _w(x):
    return ((x + (1 << 62)) % (1 << 63)) - (1 << 62)


def _idct_1d_bigint(b, col):
    W1, W2, W3, W5, W6, W7 = 2841, 2676, 2408, 1609, 1108, 565
    add, sub, mul = (lambda a, c: _w(a + c)), (lambda a, c: _w(a - c)), (lambda a, c: _w(a * c))
    asr = lambda a, k: a >> k                      # (on a value already in 63-bit range: floor, like OCaml's asr)
    if col:
        x0, x1 = add(mul(b[0], 256), 8192), mul(b[4], 256)      # lsl 8 = * 256 modulo 2^63
        r, rs, s = 4, 3, 14
    else:
        x0, x1 = add(mul(b[0], 2048), 128), mul(b[4], 2048)
        r, rs, s = 0, 0, 8
    x2, x3, x4, x5, x6, x7 = b[6], b[2], b[1], b[7], b[5], b[3]
    x8 = add(mul(W7, add(x4, x5)), r)
    x4 = asr(add(x8, mul(W1 - W7, x4)), rs)
    x5 = asr(sub(x8, mul(W1 + W7, x5)), rs)
    x8 = add(mul(W3, add(x6, x7)), r)
    x6 = asr(sub(x8, mul(W3 - W5, x6)), rs)
    x7 = asr(sub(x8, mul(W3 + W5, x7)), rs)
    x8 = add(x0, x1)
    x0 = sub(x0, x1)
    x1 = add(mul(W6, add(x3, x2)), r)
    x2 = asr(sub(x1, mul(W2 + W6, x2)), rs)
    x3 = asr(add(x1, mul(W2 - W6, x3)), rs)
    x1 = add(x4, x6)
    x4 = sub(x4, x6)
    x6 = add(x5, x7)
    x5 = sub(x5, x7)
    x7 = add(x8, x3)
    x8 = sub(x8, x3)
    x3 = add(x0, x2)
    x0 = sub(x0, x2)
    x2n = asr(add(mul(181, add(x4, x5)), 128), 8)
    x4n = asr(add(mul(181, sub(x4, x5)), 128), 8)
    x2, x4 = x2n, x4n
    return [asr(add(x7, x1), s), asr(add(x3, x2), s), asr(add(x0, x4), s), asr(add(x8, x6), s),
            asr(sub(x8, x6), s), asr(sub(x0, x4), s), asr(sub(x3, x2), s), asr(sub(x7, x1), s)]


def _decode_block_bigint(coefs_zz, q, dc_pred):
    zi = list(orc.zigzag_inverse())
    dc = _w(coefs_zz[0] + dc_pred)
    deq = [0] * 64
    deq[0] = _w(dc * q[0])
    for i in range(1, 64):
        deq[zi[i]] = _w(coefs_zz[i] * q[i])
    v = list(deq)
    for r in range(8):
        v[8 * r:8 * r + 8] = _idct_1d_bigint(v[8 * r:8 * r + 8], False)
    for c in range(8):
        col = _idct_1d_bigint([v[c + 8 * k] for k in range(8)], True)
        for k in range(8):
            v[c + 8 * k] = col[k]
    return dc, [max(-128, min(127, x)) + 128 for x in v]


def test_the_restatement_wraps_like_63_bit_ocaml_ints():
    rng = np.random.Generator(np.random.PCG64(63))
    seen_wrap = 0
    for case in range(300):
        bits = [20, 33, 40, 47, 55, 61, 62][case % 7]
        coefs = [0] * 64
        coefs[0] = int(rng.integers(-(1 << 62), 1 << 62)) >> (62 - bits)
        for k in rng.choice(np.arange(1, 64), size=int(rng.integers(0, 20)), replace=False):
            coefs[int(k)] = int(rng.integers(-1023, 1024))
        q = [int(x) for x in rng.integers(1, 65536 if case % 3 == 0 else 256, size=64)]
        dc_pred = int(rng.integers(-(1 << 62), 1 << 62)) >> int(rng.integers(0, 40))
        want_dc, want = _decode_block_bigint(coefs, q, dc_pred)
        dc, _, _, recon = orc.decode_block_summary(np.array(coefs, dtype=np.int64), np.array(q, dtype=np.int64), dc_pred)
        assert dc == want_dc, case
        assert [int(x) for x in recon] == want, case
        seen_wrap += abs(coefs[0] + dc_pred) >= (1 << 62) or abs(want_dc * q[0]) >= (1 << 62)
    assert seen_wrap > 100      # (most of these cases really leave 63 bits)
    # ... and on everything an encoder can write the wrap is the identity: the pinned vectors above are unchanged


def test_g7_packed422_kat():
    """tools/src/packed_422.ml:56-104: a 4 x 4 4:2:2 frame, packed as YUY2 and unpacked again"""
    g = golden_json("g7_packed422.json")
    A = lambda rows: np.array(rows, dtype=np.uint8)
    y, u, v = A(g["frame"][0:4]), A(g["frame"][4:8]), A(g["frame"][8:12])
    packed = orc.packed422_from_planar(orc.PACKED["YUY2"], y, u, v)
    assert packed.tolist() == g["packed"]
    y2, u2, v2 = orc.packed422_to_planar(orc.PACKED["YUY2"], packed, 4, 4)
    assert y2.tolist() + u2.tolist() + v2.tolist() == g["unpacked"]
    # the other two byte orders (packed_422.ml:6-8): the same samples at other places of each group of four bytes
    for name, (yo, uo, vo) in (("UYVY", (1, 0, 2)), ("YVYU", (0, 3, 1))):
        p = orc.packed422_from_planar(orc.PACKED[name], y, u, v).reshape(4, 2, 4)
        assert np.array_equal(p[:, :, yo], y[:, 0::2]) and np.array_equal(p[:, :, yo + 2], y[:, 1::2])
        assert np.array_equal(p[:, :, uo], u) and np.array_equal(p[:, :, vo], v)
        back = orc.packed422_to_planar(orc.PACKED[name], p.reshape(4, 8), 4, 4)
        assert all(np.array_equal(a, b) for a, b in zip(back, (y, u, v)))


def test_oconv_reproduces_the_nonstandard_size_pin():
    """test-nonstandard-sizes.t:3-15 -- `oyuv convert mini64x64.420 64x64 mini52x44.420 52x44`, then encode at quality
    95, decode, PSNR against the converted file: G4's sixth triple through the restated Oconv pipeline in one piece"""
    c = golden_json("g4_psnr_pins.json")["nonstandard"]
    small = orc.oconv_frame(golden_bytes(c["file"]), 420, (64, 64), 420, (c["width"], c["height"]))
    y, u, v = orc.split_yuv(small, c["width"], c["height"], 420)
    jpg = orc.encode_yuv(y, u, v, c["width"], c["height"], 420, c["quality"])
    got = orc.decode_a_frame(jpg)
    assert [orc.ocaml_float_to_string(orc.psnr(a, b)) for a, b in zip((y, u, v), got)] == c["psnr"]


def test_g8_code_tables_every_symbol():
    """jpeg/model/test/test_tables.ml:4-395: Tables.Encoder.dc_table / ac_table of the four default specifications, every
    symbol's length / bits / data -- create_code_table (tables.ml:27-45) and the encoder's table shaping (:504-545)."""
    g = golden_json("g8_code_tables.json")
    for name in ("dc_luma", "dc_chroma", "ac_luma", "ac_chroma"):
        assert orc.enc_table(name) == g[name], name
    assert sum(len(r) for r in g["ac_luma"]) == 176 and len(g["dc_luma"]) == 12
    # the canonical assignment itself, on a specification that is not one of the defaults (Mouse480.jpg's own DHT segments:
    # the reference prints them, test_codeblock_decoder.ml): prefix-free, lengths as specified, codes of one length consecutive
    for ht in golden_json("mouse480_header.json")["huffman_tables"]:
        codes = orc.create_code_table(ht["lengths"], ht["values"])
        assert [c[2] for c in codes] == ht["values"][:len(codes)] and len(codes) == sum(ht["lengths"])
        assert [sum(1 for c in codes if c[0] == k + 1) for k in range(16)] == ht["lengths"]
        words = [format(c[1], "0%db" % c[0]) for c in codes]
        assert len(set(words)) == len(words) and not any(a != b and b.startswith(a) for a in words for b in words)
        for a, b in zip(codes, codes[1:]):
            assert (b[1] == a[1] + 1) if a[0] == b[0] else (b[1] == (a[1] + 1) << (b[0] - a[0]))
