"""Photograph-like statistics at 1080p: every other full-size test feeds synthetic frames -- half
noise blocks, half ramps -- or the reference's two small files.  Here: the reference's Mouse480.jpg, decoded by the
oracle and enlarged to 1920 x 1080 (bicubic: smooth gradients, long runs of EOB-only blocks -- the content whose streams
fall into step slowly), as it is and with sensor-like noise on top (a larger symbol alphabet), encoded by the oracle at
three qualities, with the model's tables and with tables optimised per file; through the batch pipeline with the GPU
reader against orc.Decoder, and one by one through hvc_jpeg_decode."""
import numpy as np
import pytest

from conftest import golden_bytes
from helpers import jpeg_optimised_tables
from oracle import orc

pytestmark = pytest.mark.gpu

PIL = pytest.importorskip("PIL.Image")


def photo_frames():
    """(name, y, u, v) at 1920 x 1080 4:2:0"""
    d = orc.Decoder(golden_bytes("Mouse480.jpg"))
    d.decode()
    y, u, v = d.get_yuv_frame()
    W, H = 1920, 1080
    big = [np.asarray(PIL.fromarray(p).resize(s, PIL.BICUBIC)) for p, s in ((y, (W, H)), (u, (W // 2, H // 2)), (v, (W // 2, H // 2)))]
    rng = np.random.Generator(np.random.PCG64(42))
    noisy = [np.clip(p.astype(np.int32) + np.rint(rng.normal(0.0, s, p.shape)).astype(np.int32), 0, 255).astype(np.uint8)
             for p, s in zip(big, (4.0, 2.0, 2.0))]
    return [("smooth", *big), ("noisy", *noisy)]


def photo_files(q):
    """four 1080p files of one quality (a batch shares its quantiser tables): smooth / noisy content, the model's / per-file optimised tables"""
    W, H = 1920, 1080
    files = []
    for name, y, u, v in photo_frames():
        j = orc.encode_yuv(y, u, v, W, H, 420, q)
        files.append(("%s q%d" % (name, q), j))
        qt = np.stack([orc.quant_scale(orc.quant_luma(), q), orc.quant_scale(orc.quant_chroma(), q)])
        files.append(("%s q%d optimised tables" % (name, q), jpeg_optimised_tables(W, H, 420, qt, orc.Decoder(j).coef_record(), 2)))
    return files


@pytest.mark.parametrize("q", [50, 75, 90])
def test_photograph_like_1080p_files(q):
    import video_coding_amd as hvc
    named = photo_files(q) * 2  # (eight files: two chunks of four)
    files = [j for _, j in named]
    want = []
    for j in files[:4]:
        d = orc.Decoder(j)
        d.decode()
        want.append(np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
    want = want * 2
    fs = hvc.hvc.jpeg_read_header(files[0]).pixel_bytes
    c = hvc.Context(0)
    try:
        out = np.zeros(len(files) * fs, np.uint8)
        for gpu in (True, False):
            out[:] = 0
            st = c.jpeg_decode_batch(files, out, fs, threads=8, frames_per_chunk=4, gpu_entropy=gpu)
            for f, (name, _) in enumerate(named):
                assert np.array_equal(out[f * fs:(f + 1) * fs], want[f]), (name, "GPU reader" if gpu else "host reader")
            if gpu:  # smooth content may need many rounds: whatever does not settle is redone by the host reader, never wrong
                print("q %d: host reader time spent on chunks the GPU reader handed back: %.1f ms" % (q, st.entropy_ms_sum))
        for f, (name, j) in enumerate(named[:4]):  # the reference's own call shape: one file at a time
            _, pix = c.jpeg_decode(j)
            assert np.array_equal(np.asarray(pix).reshape(-1)[:fs], want[f]), name
    finally:
        c.close()
