"""BASELINE.json config 1's file (128x128 4:2:0, mini64x64.420 tiled 2x2 -- tests/test_config1_128.py) through the GPU path:
hvc_jpeg_encode's bytes = Encoder.encode_420's (encoder.ml:512-541), hvc_jpeg_decode's planes = Decoder.decode_a_frame's
(decoder.ml:422-427), and the PSNR lines of jpeg/test/model-encode-and-decode.t:15-17 come out digit for digit."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import config1_frame
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("quality", [75, 95, 50, 30])
def test_config1_128x128_encode_then_decode(ctx, quality):
    import video_coding_amd as hvc
    y, u, v = config1_frame()
    jpg = ctx.jpeg_encode(y, u, v, 128, 128, 420, quality)
    assert jpg == orc.encode_yuv(y, u, v, 128, 128, 420, quality)
    info, pixels = ctx.jpeg_decode(jpg)
    d = orc.Decoder(jpg)
    d.decode()
    planes = info.planes(pixels)
    for i in range(3):
        assert np.array_equal(planes[i], d.plane(i)), i
    want = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
    assert np.array_equal(hvc.hvc.jpeg_get_yuv_frame(info, pixels), want)
    if quality == 75:   # G3: the tiled decode of the reference's own mini.jpg
        _, mini = ctx.jpeg_decode(golden_bytes("mini.jpg"))
        m = orc.Decoder(golden_bytes("mini.jpg"))
        m.decode()
        for i in range(3):
            assert np.array_equal(planes[i], np.tile(m.plane(i), (2, 2))), i
    else:               # G4: the reference's printed PSNR lines
        pins = {c["quality"]: c["psnr"] for c in golden_json("g4_psnr_pins.json")["cases"] if c["file"] == "mini64x64.420"}
        got = [orc.ocaml_float_to_string(orc.psnr(a, b)) for a, b in zip((y, u, v), planes)]
        assert got == pins[quality]
