"""BASELINE.json config 1 by name: ONE 128x128 4:2:0 baseline JPEG through the model's CPU path (plumbing, no GPU).
jpeg/test_data holds no 128x128 file, so the frame is mini64x64.420 tiled 2x2 and encoded as
jpeg/test/model-encode-and-decode.t:7-17 does (Encoder.encode_420, encoder.ml:512-541).  64 is a multiple of the 16x16 MCU, so
every 8x8 block of the tiled frame is a block of the 64x64 frame: the decoded planes are the 64x64 decode tiled 2x2, and the
per-plane PSNR equals the reference's 64x64 pin digit for digit (4 x SSE over 4 x N) -- the 128x128 case is pinned by G3 and G4."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import config1_frame
from oracle import orc
from test_host_entropy import record_planes


@pytest.fixture(scope="module")
def hvc():
    import video_coding_amd as m
    m.build()
    return m.hvc


def test_oracle_decode_of_the_tiled_frame_is_the_tiled_decode_of_mini_jpg():
    y, u, v = config1_frame()
    jpg = orc.encode_yuv(y, u, v, 128, 128, 420, 75)
    d = orc.Decoder(jpg)
    assert (d.width, d.height, d.ncomp) == (128, 128, 3)
    d.decode()
    m = orc.Decoder(golden_bytes("mini.jpg"))   # G3: the model encoder's own q75 output for the 64x64 frame
    m.decode()
    for i in range(3):
        assert np.array_equal(d.plane(i), np.tile(m.plane(i), (2, 2))), i
    # the tiled file's quantised blocks are mini.jpg's, tile by tile (absolute DC)
    rec, mrec = d.coef_record(), m.coef_record()
    assert rec.size == 4 * mrec.size


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_g4_psnr_pins_hold_for_the_128x128_frame(idx):
    """model-encode-and-decode.t:15-17, 27-29, 39-41: q95 / q50 / q30 of the 4:2:0 frame"""
    c = golden_json("g4_psnr_pins.json")["cases"][idx]
    assert c["chroma"] == 420 and c["file"] == "mini64x64.420"
    y, u, v = config1_frame()
    frame = orc.decode_a_frame(orc.encode_yuv(y, u, v, 128, 128, 420, c["quality"]))
    assert [orc.ocaml_float_to_string(orc.psnr(a, b)) for a, b in zip((y, u, v), frame)] == c["psnr"]


def test_host_front_and_back_end_on_the_128x128_file(hvc):
    """the product's host half of config 1 (no GPU): header geometry, Huffman decode into coefficient records, and the
    encoder back end's bytes -- against the oracle"""
    y, u, v = config1_frame()
    want, coefs = orc.encode_yuv(y, u, v, 128, 128, 420, 75, want_coefs=True)
    info = hvc.jpeg_read_header(want)
    assert (info.width, info.height, info.n_comp) == (128, 128, 3)
    assert [(info.comp[i].decoded_width, info.comp[i].decoded_height) for i in range(3)] == [(128, 128), (64, 64), (64, 64)]
    info, rec = hvc.jpeg_entropy_decode(want)
    for got, c in zip(record_planes(info, rec), coefs):
        assert np.array_equal(got, c.reshape(got.shape))
    enc = hvc.jpeg_encoder_layout(128, 128, 420, 75)
    assert hvc.jpeg_entropy_encode(enc, np.concatenate([c.reshape(-1) for c in coefs])) == want
