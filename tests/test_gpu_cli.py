"""The reference's cram sessions (jpeg/test/model-encode-and-decode.t, test-nonstandard-sizes.t) run
through this repository's command line -- `model encode frame`, `model decode frame`,
`oyuv compare psnr` -- with every pixel on the GPU path; the printed lines equal the pinned ones."""
import numpy as np
import pytest

import pathlib

from conftest import GOLDEN as _GOLDEN, golden_bytes, golden_json
from oracle import orc

GOLDEN = pathlib.Path(_GOLDEN)

pytestmark = pytest.mark.gpu


def cli(*argv):
    from video_coding_amd.__main__ import main
    main([str(a) for a in argv])


@pytest.mark.parametrize("idx", range(5))
def test_model_encode_and_decode_cram(tmp_path, capsys, idx):
    c = golden_json("g4_psnr_pins.json")["cases"][idx]
    src, size = GOLDEN / c["file"], "%dx%d" % (c["width"], c["height"])
    jpg, out = tmp_path / "model.jpg", tmp_path / "out_model.yuv"
    cli("model", "encode", "frame", src, size, jpg, "-quality", c["quality"], "-chroma", c["chroma"])
    cli("model", "decode", "frame", jpg, out)
    capsys.readouterr()
    cli("oyuv", "compare", "psnr", "yuv", src, out, size, "-format", c["chroma"])
    assert capsys.readouterr().out.split() == c["psnr"]
    # the file the CLI wrote is the model's, byte for byte
    y, u, v = orc.split_yuv(golden_bytes(c["file"]), c["width"], c["height"], c["chroma"])
    assert jpg.read_bytes() == orc.encode_yuv(y, u, v, c["width"], c["height"], c["chroma"], c["quality"])


def test_nonstandard_size_cram(tmp_path, capsys):
    """test-nonstandard-sizes.t, every step through this repository's command line: `oyuv convert` 64x64 -> 52x44
    (supersample_hv2, Yuv.crop, subsample_hv2 on the GPU), `model encode frame`, `model decode frame`, `oyuv compare psnr`
    -- the reference's three printed lines (the ffmpeg steps of the session have no counterpart here)."""
    c = golden_json("g4_psnr_pins.json")["nonstandard"]
    src = tmp_path / "mini52x44.420"
    for _ in range(2):   # (the session runs the conversion twice)
        cli("oyuv", "convert", GOLDEN / c["file"], "64x64", src, "52x44")
    jpg, out = tmp_path / "model.jpg", tmp_path / "out_model.yuv"
    cli("model", "encode", "frame", src, "52x44", jpg, "-quality", c["quality"])
    cli("model", "decode", "frame", jpg, out)
    capsys.readouterr()
    cli("oyuv", "compare", "psnr", "yuv", src, out, "52x44")
    assert capsys.readouterr().out.split() == c["psnr"]
    # (afterwards, as a second witness: the converted file is the restated tool's)
    assert src.read_bytes() == orc.oconv_frame(golden_bytes(c["file"]), 420, (64, 64), 420, (52, 44))


def test_convert_frames_formats_and_offsets(tmp_path):
    """`oyuv convert` with -frames, -format / -out-format (packed too) and -src-offset over a three-frame file"""
    rng = np.random.Generator(np.random.PCG64(9))
    frames = [rng.integers(0, 256, size=48 * 32 * 3 // 2, dtype=np.uint8).tobytes() for _ in range(3)]
    src, dst = tmp_path / "in.yuv", tmp_path / "out.yuv"
    src.write_bytes(b"".join(frames))
    cli("oyuv", "convert", src, "48x32", dst, "40x24", "-frames", "1-2", "-out-format", "uyvy", "-src-offset", "3,5")
    want = b"".join(orc.oconv_frame(f, 420, (48, 32), "UYVY", (40, 24), (3, 5)) for f in frames[1:3])
    assert dst.read_bytes() == want
    cli("oyuv", "convert", src, "48x32", dst, "-frames", "2-7", "-out-format", "422")       # past the end: one frame comes out
    assert dst.read_bytes() == orc.oconv_frame(frames[2], 420, (48, 32), 422, (48, 32))
    cli("oyuv", "convert", src, "48x32", dst)                                                # defaults: frame 0, same format and size
    assert dst.read_bytes() == frames[0][:48 * 32] + orc.oconv_frame(frames[0], 420, (48, 32), 420, (48, 32))[48 * 32:]


def test_convert_420_to_444_and_decode_to_444(tmp_path):
    """`oyuv convert` 4:2:0 -> 4:4:4 (K2) of the decoded frame == `model decode frame -yuv444` (fused)."""
    jpg = GOLDEN / "Mouse480.jpg"
    a, b, c = tmp_path / "a.yuv", tmp_path / "a444.yuv", tmp_path / "b444.yuv"
    cli("model", "decode", "frame", jpg, a)
    cli("oyuv", "convert", a, "480x320", b, "-format", 420, "-out-format", 444)
    cli("model", "decode", "frame", jpg, c, "-yuv444")
    assert b.read_bytes() == c.read_bytes()
    d = orc.Decoder(golden_bytes("Mouse480.jpg"))
    d.decode()
    y, u, v = d.get_yuv_frame()
    want = y.tobytes() + orc.supersample_hv2(u).tobytes() + orc.supersample_hv2(v).tobytes()
    assert c.read_bytes() == want
