"""The restart-interval extension (tests/test_restart_intervals.py) through the file-level entry points on the GPU: with
hvc_set_restart_markers on, a file with DRI / RSTn decodes to the pixels of the same frame written without them -- one file at
a time (small, and large enough for the GPU reader to be tried: it hands such files to the host reader), fused 4:4:4, both
batch pipelines; off (the default), it decodes as the model decodes it."""
import numpy as np
import pytest

from helpers import jpeg_optimised_tables
from oracle import orc
from test_restart_intervals import QT, random_record

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def planes_of(jpg):
    d = orc.Decoder(jpg)
    d.decode()
    return d


@pytest.mark.parametrize("w,h,ri", [(96, 64, 1), (200, 72, 5), (1920, 1088, 120), (1920, 1088, 7)])
def test_files_to_pixels(ctx, w, h, ri):
    import video_coding_amd as hvc
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], w, h, w + ri)
    plain = jpeg_optimised_tables(w, h, 420, QT, rec)
    marked = jpeg_optimised_tables(w, h, 420, QT, rec, restart_interval=ri)
    want = planes_of(plain)
    info, pixels = ctx.jpeg_decode(marked)                       # default: the model's reading of the marked file
    try:
        model = planes_of(marked)
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, model.plane(i)), i
    except ValueError:
        pass
    ctx.set_restart_markers(True)
    info, pixels = ctx.jpeg_decode(marked)
    for i, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, want.plane(i)), (w, h, ri, i)
    _, frame = ctx.jpeg_decode_yuv444(marked)
    y, u, v = want.get_yuv_frame()
    assert np.array_equal(frame[0], y) and np.array_equal(frame[1], orc.supersample_hv2(u)) and np.array_equal(frame[2], orc.supersample_hv2(v))
    _, pixels = ctx.jpeg_decode(plain)                           # a file without DRI: the same either way
    for i, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, want.plane(i))
    ctx.set_restart_markers(False)
    assert len(marked) > 128 * 1024 or w < 1000                 # (the large ones pass the size where the GPU reader is tried)


@pytest.mark.parametrize("gpu_entropy", [False, True])
def test_batches(ctx, gpu_entropy):
    import video_coding_amd as hvc
    w, h = 328, 200
    recs = [random_record([(2, 2), (1, 1), (1, 1)], w, h, 40 + f)[0] for f in range(9)]
    marked = [jpeg_optimised_tables(w, h, 420, QT, r, restart_interval=3 + f % 4) for f, r in enumerate(recs)]
    wants = [planes_of(jpeg_optimised_tables(w, h, 420, QT, r)) for r in recs]
    info = hvc.hvc.jpeg_read_header(marked[0])
    fs = info.pixel_bytes
    ctx.set_restart_markers(True)
    out = np.zeros(len(marked) * fs, np.uint8)
    ctx.jpeg_decode_batch(marked, out, fs, threads=3, frames_per_chunk=4, gpu_entropy=gpu_entropy)
    for f in range(len(marked)):
        for i, plane in enumerate(info.planes(out[f * fs:(f + 1) * fs])):
            assert np.array_equal(plane, wants[f].plane(i)), (f, i)
    info_r, recs_gpu, used = ctx.jpeg_entropy_decode_gpu(marked[:3], device=False)
    assert used == 0 and all(np.array_equal(recs_gpu[f], recs[f]) for f in range(3))   # handed to the host reader, which honours the markers
