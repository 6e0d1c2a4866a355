"""G9: the reference's cram tests compare the model's decoder with ffmpeg and accept a maximum difference
of 1 per sample (jpeg/test/mouse-decode.t:10-13: 1 / 0 / 0 for Mouse480; model-encode-and-decode.t:10-13).  ffmpeg is
not in this image; libjpeg-turbo is, behind PIL -- an implementation of the same standard that shares no code with the
model or with this repository.  A tolerance check only: it cannot pin a bit, but it would catch an oracle (and with it
every parity test) that had drifted from what a JPEG decoder is.  CPU only; skipped where PIL is missing.

libjpeg hands out chroma only after its own ("fancy") upsampling, so 4:2:0 files are compared on the luma plane and the
three planes are compared on 4:4:4 files, where nothing is resampled."""
import io

import numpy as np
import pytest

from conftest import golden_bytes
from helpers import synth_pixels
from oracle import orc

Image = pytest.importorskip("PIL.Image")


def libjpeg_ycc(jpeg):
    im = Image.open(io.BytesIO(jpeg))
    im.draft("YCbCr", im.size)  # the decoder's own colour space: no RGB round trip
    im.load()
    assert im.mode == "YCbCr"
    return np.asarray(im).astype(np.int64)


def model_planes(jpeg):
    d = orc.Decoder(jpeg)
    d.decode()
    return [np.asarray(d.cropped_plane(i)).astype(np.int64) for i in range(d.ncomp)]


@pytest.mark.parametrize("name", ["Mouse480.jpg", "mini.jpg"])
def test_reference_files_luma_within_one_of_libjpeg(name):
    jpeg = golden_bytes(name)
    got, want = model_planes(jpeg)[0], libjpeg_ycc(jpeg)[..., 0]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1  # the reference's own bound against ffmpeg
    assert (got != want).mean() < 0.02   # and the two agree on all but a per cent of the samples


@pytest.mark.parametrize("w,h,q", [(64, 48, 75), (200, 120, 95), (96, 64, 30)])
def test_444_files_all_planes_within_one_of_libjpeg(w, h, q):
    r8 = lambda x: (x + 7) // 8 * 8
    planes = [synth_pixels(40 + i, r8(h), r8(w))[:h, :w] for i in range(3)]
    jpeg = orc.encode_yuv(planes[0], planes[1], planes[2], w, h, 444, q)
    want = libjpeg_ycc(jpeg)
    for i, got in enumerate(model_planes(jpeg)):
        assert got.shape == want[..., i].shape
        assert np.abs(got - want[..., i]).max() <= 1, "component %d" % i
