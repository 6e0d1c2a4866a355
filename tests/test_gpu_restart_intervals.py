"""The restart-interval extension (tests/test_restart_intervals.py) through the file-level entry points on the GPU: with
hvc_set_restart_markers on, a file with DRI / RSTn decodes to the pixels of the same frame written without them -- one file at
a time (small, and large enough for the GPU reader to be tried), fused 4:4:4, both batch pipelines; off (the default), it
decodes as the model decodes it.  The GPU Huffman reader takes every interval as a frame of its own (hvc_hdec.h
HdParams::rst_*): its records equal the host reader's, with `used_gpu` / no host-reader time saying that it was the GPU's."""
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
    assert used == 0 and all(np.array_equal(recs_gpu[f], recs[f]) for f in range(3))   # files of different intervals: the host reader's, which honours the markers


SAMPLINGS = [[(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 1)], [(1, 2), (1, 1), (1, 1)]]


@pytest.mark.parametrize("sampling", SAMPLINGS)
@pytest.mark.parametrize("device", [False, True])
def test_the_gpu_reader_takes_intervals_as_frames(ctx, sampling, device):
    """records of the GPU reader = the records the files were written from, for intervals of one MCU, a few, a row of MCUs,
    all but one, and with the last interval shorter than the others; per-file (optimised) and shared tables"""
    ctx.set_restart_markers(True)
    for (w, h) in ((64, 48), (200, 72), (328, 200)):
        recs, n_mcu = zip(*[random_record(sampling, w, h, 7 * w + f) for f in range(3)])
        n_mcu = n_mcu[0]
        mh = max(s[0] for s in sampling)
        row = -(-w // (8 * mh))
        for ri in sorted({1, 3, row, 2 * row + 1, n_mcu - 1}):
            if ri < 1 or ri >= n_mcu:
                continue
            for sets in (1, min(2, len(sampling))):
                files = [jpeg_optimised_tables(w, h, sampling, QT, r, table_sets=sets, restart_interval=ri) for r in recs]
                for batch in (files, files[:1]):   # (three files = three sets of tables: per-frame tables; one: the LDS-table kernels)
                    info, got, used = ctx.jpeg_entropy_decode_gpu(batch, device=device)
                    assert used == 1, (w, h, ri, sets, len(batch))
                    for f in range(len(batch)):
                        assert np.array_equal(np.asarray(got[f])[:recs[f].size], recs[f]), (w, h, ri, sets, f)
    ctx.set_restart_markers(False)


def test_the_gpu_reader_off_by_default_and_on_unusual_streams(ctx):
    """default: a marked file is the model's (the segment ends at the first RSTn: the reader hands the truncated stream to
    the host reader); on: a file with fewer / more markers than its DRI promises, or an interval cut short, is the host
    reader's -- same records either way"""
    import video_coding_amd as hvc
    w, h, ri = 200, 72, 5
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], w, h, 99)
    good = jpeg_optimised_tables(w, h, 420, QT, rec, restart_interval=ri)
    try:                                     # (default: the model's reading -- its DC predictors run on over the first interval's end)
        _, want = hvc.hvc.jpeg_entropy_decode(good)
        info, got, used = ctx.jpeg_entropy_decode_gpu([good])
        assert used == 0 and np.array_equal(got[0], want)
    except hvc.HvcError as e:
        with pytest.raises(hvc.HvcError) as e2:
            ctx.jpeg_entropy_decode_gpu([good])
        assert e2.value.code == e.code
    ctx.set_restart_markers(True)
    b = bytearray(good)
    marks = [i for i in range(len(b) - 1) if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7]
    assert len(marks) == -(-n_mcu // ri) - 1
    variants = {"one marker less": bytes(b[:marks[3]] + b[marks[3] + 2:]),
                "one marker more": bytes(b[:marks[2]] + b[marks[2]:marks[2] + 2] + b[marks[2]:]),
                "an interval cut short": bytes(b[:marks[4] - 9] + b[marks[4]:]),
                "a fill byte in front of a marker": bytes(b[:marks[1]] + b"\xff" + b[marks[1]:])}
    for name, f in variants.items():
        try:
            _, want = hvc.hvc.jpeg_entropy_decode(f, restart_markers=True)
        except hvc.HvcError as e:
            with pytest.raises(hvc.HvcError) as e2:
                ctx.jpeg_entropy_decode_gpu([f])
            assert e2.value.code == e.code, name
            continue
        info, got, used = ctx.jpeg_entropy_decode_gpu([f])
        assert np.array_equal(got[0], want), name
        assert used == (1 if name == "a fill byte in front of a marker" else used), name
    ctx.set_restart_markers(False)


@pytest.mark.parametrize("yuv444", [False, True])
def test_the_gpu_pipeline_with_intervals(ctx, yuv444):
    """hvc_jpeg_decode_batch_gpu: 1080p files with a row of MCUs per interval (what encoders write), shared and per-file
    tables in one batch; no chunk falls to the host reader"""
    import video_coding_amd as hvc
    w, h, ri, n = 1920, 1080, 120, 10
    recs = [random_record([(2, 2), (1, 1), (1, 1)], w, h, 300 + f)[0] for f in range(3)]
    marked = [jpeg_optimised_tables(w, h, 420, QT, r, restart_interval=ri) for r in recs]
    wants = [planes_of(jpeg_optimised_tables(w, h, 420, QT, r)) for r in recs]
    info = hvc.hvc.jpeg_read_header(marked[0])
    fs = 3 * w * h if yuv444 else info.pixel_bytes
    ctx.set_restart_markers(True)
    for which in (lambda f: 0, lambda f: f % 3):   # one file ten times (one set of tables), three files in turn (tables per file)
      files = [marked[which(f)] for f in range(n)]
      out = np.zeros(n * fs, np.uint8)
      st = ctx.jpeg_decode_batch(files, out, fs, threads=4, frames_per_chunk=4, gpu_entropy=True, yuv444=yuv444)
      assert st.entropy_ms_sum == 0.0          # (host-reader time: only chunks handed back have any)
      for f in range(n):
        d = wants[which(f)]
        if yuv444:
            y, u, v = d.get_yuv_frame()
            fr = out[f * fs:(f + 1) * fs].reshape(3, h, w)
            assert np.array_equal(fr[0], y) and np.array_equal(fr[1], orc.supersample_hv2(u)) and np.array_equal(fr[2], orc.supersample_hv2(v)), f
        else:
            for i, plane in enumerate(info.planes(out[f * fs:(f + 1) * fs])):
                assert np.array_equal(plane, d.plane(i)), (f, i)
    ctx.set_restart_markers(False)
