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
