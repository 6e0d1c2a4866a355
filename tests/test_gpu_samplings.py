"""Sampling factors the model's encoder never writes but its decoder reads (Decoder.init / decode_seq, decoder.ml:294-345,
362-395): 4:4:0, 4:1:1, a first component that is not the largest, factors of three, one / two / four components -- whole
files to pixels through the C ABI (host reader or GPU reader + the block stage), one at a time and through both batch
pipelines, against the model restatement."""
import numpy as np
import pytest

from oracle import orc
from test_host_entropy import UNUSUAL_SAMPLINGS, unusual_sampling_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def model_planes(jpg):
    d = orc.Decoder(jpg)
    d.decode()
    return d, [d.plane(i) for i in range(d.ncomp)]


@pytest.mark.parametrize("si", range(len(UNUSUAL_SAMPLINGS)))
def test_one_file_at_a_time(ctx, si):
    import video_coding_amd as hvc
    for (w, h, seed) in ((40, 24, 1), (97, 51, 2), (640, 360, 3)):   # (the last one is large enough for the GPU reader)
        jpg, _ = unusual_sampling_file(UNUSUAL_SAMPLINGS[si], w, h, 1000 * si + seed)
        info, pixels = ctx.jpeg_decode(jpg)
        d, want = model_planes(jpg)
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, want[i]), (si, w, i)
        crop = np.concatenate([p.reshape(-1) for p in d.cropped_planes()])
        assert np.array_equal(hvc.hvc.jpeg_get_cropped_planes(info, pixels), crop)
        # Decoder.get_yuv_frame: Frame.of_planes has no name for these planes (but for none of them... it raises)
        try:
            frame = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
        except ValueError:
            frame = None
        try:
            got = hvc.hvc.jpeg_get_yuv_frame(info, pixels)
        except hvc.HvcError as e:
            assert e.code == -8
            got = None
        assert (got is None) == (frame is None) and (got is None or np.array_equal(got, frame))


@pytest.mark.parametrize("gpu_entropy", [False, True])
@pytest.mark.parametrize("si", range(len(UNUSUAL_SAMPLINGS)))
def test_batches(ctx, si, gpu_entropy):
    """eleven files of one sampling (different content, the first file's tables differ from the others': every file is
    written with the tables optimal for itself) through hvc_jpeg_decode_batch / hvc_jpeg_decode_batch_gpu; a sampling the
    GPU reader does not take (more than three components, more than 16 blocks per MCU) goes through the host reader there"""
    import video_coding_amd as hvc
    jpegs = [unusual_sampling_file(UNUSUAL_SAMPLINGS[si], 328, 200, 5000 + 100 * si + f)[0] for f in range(11)]
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    stride = info.pixel_bytes
    pixels = np.zeros(len(jpegs) * stride, dtype=np.uint8)
    ctx.jpeg_decode_batch(jpegs, pixels, stride, threads=3, frames_per_chunk=4, gpu_entropy=gpu_entropy)
    for f, j in enumerate(jpegs):
        _, want = model_planes(j)
        for i, plane in enumerate(info.planes(pixels[f * stride:(f + 1) * stride])):
            assert np.array_equal(plane, want[i]), (si, f, i)


def test_the_gpu_reader_takes_what_it_can(ctx):
    """hvc_jpeg_entropy_decode_gpu on these samplings: records equal to the host reader's whether the GPU reader took the
    batch (used = 1: at most three components and 16 blocks per MCU) or handed it back (used = 0: nothing to compare)"""
    import video_coding_amd as hvc
    took = 0
    for si, sampling in enumerate(UNUSUAL_SAMPLINGS):
        jpegs = [unusual_sampling_file(sampling, 640, 360, 9000 + 10 * si + f)[0] for f in range(3)]
        info, recs, used = ctx.jpeg_entropy_decode_gpu(jpegs, device=True)
        blocks_per_mcu = sum(a * b for a, b in sampling)
        if len(sampling) > 3 or blocks_per_mcu > 16:
            assert used == 0, sampling
        if used:
            took += 1
            for f, j in enumerate(jpegs):
                assert np.array_equal(recs[f], hvc.hvc.jpeg_entropy_decode(j)[1]), (si, f)
    assert took >= 8
