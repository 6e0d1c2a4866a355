"""BASELINE config 3 at its real shape inside the suite: 200 x 1080p 4:2:0 baseline JPEG files through
hvc_jpeg_decode_batch (host Huffman reader || H2D || block stage) and hvc_jpeg_decode_batch_gpu (Huffman
reader on the GPU), frames_per_chunk = 0 -- the library's own chunk rule, several chunks, every ring slot
used more than once -- with the frames left in device memory and delivered to host memory.  The files are the
MODEL's (orc.encode_yuv = Encoder.encode_420); every decoded frame must equal orc.Decoder's planes
(Decoder.decode, jpeg/model/src/decoder.ml:362-427) byte for byte, and its K5 checksum the numpy one."""
import numpy as np
import pytest

from helpers import checksum_records, synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
N_FILES = 200
N_DISTINCT = 4


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def files():
    """(jpegs in batch order, index of each one's distinct frame, the model's padded pixel records, the
    model's fused 4:4:4 frames)"""
    jpegs, padded, fused = [], [], []
    for f in range(N_DISTINCT):  # one quality = one set of quantiser tables per batch; four different contents / file sizes
        y = synth_pixels(1000 + f, 1088, 1920)[:H]
        u = synth_pixels(2000 + f, 544, 960)[:H // 2]
        v = synth_pixels(3000 + f, 544, 960)[:H // 2]
        if f == 1:
            y = np.clip(y.astype(np.int32) // 2 + 60, 0, 255).astype(np.uint8)
        if f == 3:  # a smooth frame: short blocks, many per subsequence, slow to synchronise
            y = (np.add.outer(np.arange(H), np.arange(W)) // 9 % 256).astype(np.uint8)
        jpegs.append(orc.encode_yuv(y, u, v, W, H, 420, 75))
    for j in jpegs:
        d = orc.Decoder(j)
        d.decode()
        padded.append(np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
        yy, uu, vv = d.get_yuv_frame()
        fused.append(np.concatenate([yy.reshape(-1), orc.supersample_hv2(uu).reshape(-1), orc.supersample_hv2(vv).reshape(-1)]))
    rng = np.random.Generator(np.random.PCG64(33))
    order = rng.integers(0, N_DISTINCT, size=N_FILES)  # no period: chunk boundaries fall anywhere
    order[:N_DISTINCT] = np.arange(N_DISTINCT)
    return [jpegs[k] for k in order], order, padded, fused


@pytest.mark.parametrize("gpu_entropy", [False, True])
@pytest.mark.parametrize("device_out", [True, False])
def test_config3_shape_default_chunking(ctx, files, gpu_entropy, device_out):
    import torch
    import video_coding_amd as hvc
    batch, order, padded, _ = files
    info = hvc.hvc.jpeg_read_header(batch[0])
    fs = info.pixel_bytes
    assert fs == padded[0].size == 1920 * 1088 * 3 // 2
    if device_out:
        out = torch.zeros(N_FILES * fs, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
    else:
        out = np.zeros(N_FILES * fs, dtype=np.uint8)
    for rep in range(2):  # the second call finds the rings allocated
        st = ctx.jpeg_decode_batch(batch, out, fs, threads=8, frames_per_chunk=0, gpu_entropy=gpu_entropy)
        assert st.chunks >= 4, "every ring slot (3) must come round again"
        assert st.frames_per_chunk == (64 if gpu_entropy else 32)  # include/hvc_jpeg.h: the default chunk rule
        if gpu_entropy:
            assert st.entropy_ms_sum == 0, "a chunk fell to the host reader"
        sums = ctx.checksum_records(out, fs, N_FILES)
        got = out.cpu().numpy() if device_out else out
        got = got.reshape(N_FILES, fs)
        want_sums = checksum_records(np.stack(padded))
        for f in range(N_FILES):
            assert np.array_equal(got[f], padded[order[f]]), (rep, f)
            assert sums[f] == want_sums[order[f]], (rep, f)
        if device_out:
            out.zero_()
            torch.cuda.synchronize()
        else:
            out[:] = 0


def test_config3_shape_fused_444_output(ctx, files):
    """the same batch through the fused block stage (4:2:0 files -> tight 4:4:4 frames), GPU reader, default chunks"""
    import torch
    batch, order, _, fused = files
    fs = 3 * W * H
    out = torch.zeros(N_FILES * fs, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = ctx.jpeg_decode_batch(batch, out, fs, threads=8, frames_per_chunk=0, yuv444=True, gpu_entropy=True)
    assert st.chunks >= 4 and st.entropy_ms_sum == 0
    sums = ctx.checksum_records(out, fs, N_FILES)
    want_sums = checksum_records(np.stack(fused))
    assert all(sums[f] == want_sums[order[f]] for f in range(N_FILES))
    got = out.cpu().numpy().reshape(N_FILES, fs)
    for f in range(0, N_FILES, 7):
        assert np.array_equal(got[f], fused[order[f]]), f


def test_config3_shape_with_per_file_tables(ctx, files):
    """The same shape with the files' Huffman tables re-written: every file its own optimised tables, two or three
    different sets per file, and files with the model's tables in between -- chunks in PF mode with one work list per
    frame (k_hd_sync_pf), tables in LDS for some frames and in device memory for others, next to chunks that happen to
    be uniform.  Same coefficients, so the model's frames are the reference as before."""
    import torch
    import video_coding_amd as hvc
    from helpers import jpeg_optimised_tables
    batch, order, padded, _ = files
    qt = np.stack([orc.quant_scale(orc.quant_luma(), 75), orc.quant_scale(orc.quant_chroma(), 75)])
    variants = {}
    for k in range(N_DISTINCT):
        rec = orc.Decoder(batch[k]).coef_record()   # batch[:N_DISTINCT] are the distinct files in order
        variants[k] = [batch[k], jpeg_optimised_tables(W, H, 420, qt, rec, 2), jpeg_optimised_tables(W, H, 420, qt, rec, 3)]
    rng = np.random.Generator(np.random.PCG64(77))
    n = 96
    pick = rng.integers(0, 3, size=n)
    pick[70:] = 0                                   # the last chunk(s): the model's tables throughout
    mixed = [variants[order[f]][pick[f]] for f in range(n)]
    info = hvc.hvc.jpeg_read_header(mixed[0])
    fs = info.pixel_bytes
    out = torch.zeros(n * fs, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    st = ctx.jpeg_decode_batch(mixed, out, fs, threads=8, frames_per_chunk=16, gpu_entropy=True)
    assert st.chunks == 6 and st.entropy_ms_sum == 0, "a chunk fell to the host reader"
    sums = ctx.checksum_records(out, fs, n)
    want_sums = checksum_records(np.stack(padded))
    got = out.cpu().numpy().reshape(n, fs)
    for f in range(n):
        assert sums[f] == want_sums[order[f]], f
        assert np.array_equal(got[f], padded[order[f]]), f
