"""The two corners of tests/test_model_corners.py, files to pixels on the GPU through the C ABI: components without blocks
(the model's empty planes) and DC categories of 33 ... 62 bits (the model's 63-bit arithmetic, wrap-around included) --
one file at a time, through both batch pipelines, and through the fused 4:4:4 output, against the model restatement."""
import numpy as np
import pytest

from conftest import golden_bytes
from helpers import jpeg_optimised_tables
from oracle import orc
from test_model_corners import EMPTY_PLANE_SAMPLINGS, empty_plane_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def model(jpg):
    d = orc.Decoder(jpg)
    d.decode()
    return d


@pytest.mark.parametrize("si", range(len(EMPTY_PLANE_SAMPLINGS)))
def test_files_with_an_empty_plane_decode_like_the_model(ctx, si):
    import video_coding_amd as hvc
    sampling = EMPTY_PLANE_SAMPLINGS[si]
    for (w, h, seed) in ((40, 24, 1), (97, 51, 2), (640, 360, 3)):   # (the last one: past the size where the GPU reader is tried)
        jpg, _ = empty_plane_file(sampling, w, h, 77 * si + seed)
        d = model(jpg)
        info, pixels = ctx.jpeg_decode(jpg)
        assert info.pixel_bytes == sum(d.plane(i).size for i in range(d.ncomp))
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, d.plane(i)), (sampling, w, i)
        with pytest.raises(hvc.HvcError) as e:                         # Frame.of_planes raises
            hvc.hvc.jpeg_get_yuv_frame(info, pixels)
        assert e.value.code == -8
        assert np.array_equal(hvc.hvc.jpeg_get_cropped_planes(info, pixels),
                              np.concatenate([p.reshape(-1) for p in d.cropped_planes()]))


@pytest.mark.parametrize("gpu_entropy", [False, True])
@pytest.mark.parametrize("si", [0, 2, 5])
def test_batches_of_files_with_an_empty_plane(ctx, si, gpu_entropy):
    import video_coding_amd as hvc
    jpegs = [empty_plane_file(EMPTY_PLANE_SAMPLINGS[si], 328, 200, 3000 + 10 * si + f)[0] for f in range(9)]
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    stride = info.pixel_bytes
    pixels = np.zeros(len(jpegs) * stride, dtype=np.uint8)
    ctx.jpeg_decode_batch(jpegs, pixels, stride, threads=3, frames_per_chunk=4, gpu_entropy=gpu_entropy)
    for f, j in enumerate(jpegs):
        d = model(j)
        for i, plane in enumerate(info.planes(pixels[f * stride:(f + 1) * stride])):
            assert np.array_equal(plane, d.plane(i)), (si, f, i)


def test_the_record_level_block_stage_skips_components_without_blocks(ctx):
    """hvc_decode_frames with the layout hvc_jpeg_read_header reports for such a file: the empty component takes no part,
    the others land where their offsets say (device memory and host memory)"""
    import torch
    import video_coding_amd as hvc
    jpg, rec = empty_plane_file(EMPTY_PLANE_SAMPLINGS[0], 200, 120, 5)
    d = model(jpg)
    info = hvc.hvc.jpeg_read_header(jpg)
    comps = [dict(blocks_w=info.layout[i].blocks_w, blocks_h=info.layout[i].blocks_h, qtab=info.layout[i].qtab,
                  coef_offset=info.layout[i].coef_offset, plane_offset=info.layout[i].plane_offset, stride=info.layout[i].stride)
             for i in range(info.n_comp)]
    assert any(c["blocks_w"] * c["blocks_h"] == 0 for c in comps)
    want = np.concatenate([d.plane(i).reshape(-1) for i in range(d.ncomp)])
    out = np.zeros(info.pixel_bytes, dtype=np.uint8)
    ctx.decode_frames(rec, info.coef_count, info.qtab_array(), comps, 1, out, info.pixel_bytes)
    assert np.array_equal(out, want)
    d_rec = torch.from_numpy(rec.copy()).cuda()
    d_out = torch.zeros(info.pixel_bytes, dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.decode_frames(d_rec, info.coef_count, info.qtab_array(), comps, 1, d_out, info.pixel_bytes)
    ctx.synchronize()
    ctx.reset_stream()
    assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.parametrize("which", ["width", "height"])
def test_frames_without_width_or_height(ctx, which):
    """no MCU, no block, planes without a sample: every file-level entry point returns as Decoder.decode does"""
    import video_coding_amd as hvc
    data = bytearray(golden_bytes("mini.jpg"))
    sof = bytes(data).index(b"\xff\xc0")
    at = sof + 7 if which == "width" else sof + 5
    data[at:at + 2] = b"\0\0"
    data = bytes(data)
    model(data)
    info, pixels = ctx.jpeg_decode(data)
    assert info.pixel_bytes == 0 and pixels.size == 0
    assert hvc.hvc.jpeg_get_yuv_frame(info, np.zeros(0, np.uint8)).size == 0
    _, frame = ctx.jpeg_decode_yuv444(data)
    assert frame.size == 0
    for gpu in (False, True):
        out = np.zeros(8, dtype=np.uint8)
        ctx.jpeg_decode_batch([data] * 5, out, 0, threads=2, frames_per_chunk=2, gpu_entropy=gpu)
        assert not out.any()


# ---------------------------------------------------------------------------------------------------------------------

def wide_dc_file_420(cats, w, h, q0, seed):
    """a 4:2:0 file: every component's DC differences take the given categories in turn (random magnitudes and signs).
    cats = "wrap": absolute DCs of m * 2^56 + s with small s, to be used with q0 = 128 -- the dequantised DC is
    m * 2^63 + 128 s, which the model's 63-bit multiplication turns into 128 s: blocks of ordinary grey levels whose every
    bit depends on the wrap-around (differences of 57 ... 62 bits)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    Wr, Hr = -(-w // 16) * 16, -(-h // 16) * 16
    nbs = [(Wr // 8) * (Hr // 8), (Wr // 16) * (Hr // 16), (Wr // 16) * (Hr // 16)]
    rec = np.zeros((sum(nbs), 64), dtype=object)
    at = 0
    for ci, nb in enumerate(nbs):
        bw = Wr // 8 if ci == 0 else Wr // 16
        # the blocks of the component in SCAN order (decode_seq, decoder.ml:374-395): the writer codes the differences of
        # blocks that follow each other there, and every one of them has to have its chosen category
        if ci == 0:
            order = [(2 * my + sy) * bw + 2 * mx + sx for my in range(Hr // 16) for mx in range(Wr // 16) for sy in range(2) for sx in range(2)]
        else:
            order = list(range(nb))
        acc = 0
        for i, b in enumerate(order):
            if cats == "wrap":
                acc = (int(rng.integers(-15, 16)) << 56) + int(rng.integers(-7, 8))
            else:
                c = cats[(i + ci) % len(cats)]
                if c:
                    mag = (1 << (c - 1)) | (int(rng.integers(0, 1 << 62)) & ((1 << (c - 1)) - 1))
                    acc += mag if rng.integers(0, 2) else -mag
            rec[at + b, 0] = acc
            for k in rng.choice(np.arange(1, 64), size=int(rng.integers(0, 6)), replace=False):
                rec[at + b, int(k)] = int(rng.integers(-40, 41))
        at += nb
    qt = np.stack([np.concatenate([[q0], np.arange(2, 65)]), np.concatenate([[q0], np.arange(64, 1, -1)])]).astype(np.uint16)
    return jpeg_optimised_tables(w, h, 420, qt, rec.reshape(-1), table_sets=2)


@pytest.mark.parametrize("cats,q0", [([33, 35, 40, 0, 47], 1), ([48, 55, 61, 62], 1), ([62, 62, 62], 255), ([20, 33, 11, 62, 47, 3], 97),
                                     ([40, 41, 42, 43], 2), ("wrap", 128)])
def test_dc_categories_up_to_62_bits(ctx, cats, q0):
    """decoder.ml:81-96 reads whatever category the table names; from there on the model computes modulo 2^63.  The
    files-to-pixels entry points carry such DCs on the side list through the int64 fix-up, which reads its sums and
    products as 63-bit numbers exactly where the model looks at them (csrc/hvc_kernels.hip idct_1d_wide): the model's
    planes, one file at a time, fused 4:4:4, and both batch pipelines."""
    import torch
    import video_coding_amd as hvc
    files = [wide_dc_file_420(cats, 72, 40, q0, (10 * sum(cats) if cats != "wrap" else 5) + k) for k in range(3)]
    wants = [model(j) for j in files]
    if cats == "wrap":   # (the point of this case: not saturated planes but grey levels made by the wrap-around)
        assert all(len(np.unique(d.plane(0))) > 20 for d in wants)
    assert max(int(np.abs(orc.Decoder(j).coef_record().astype(np.float64)).max()) for j in files) > 2.0 ** 31
    info = hvc.hvc.jpeg_read_header(files[0])
    for j, d in zip(files, wants):
        with pytest.raises(hvc.HvcError) as e:
            hvc.hvc.jpeg_entropy_decode(j)
        assert e.value.code == -5
        _, pixels = ctx.jpeg_decode(j)
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, d.plane(i)), i
        _, frame = ctx.jpeg_decode_yuv444(j)
        y, u, v = d.get_yuv_frame()
        assert np.array_equal(frame[0], y) and np.array_equal(frame[1], orc.supersample_hv2(u)) and \
            np.array_equal(frame[2], orc.supersample_hv2(v))
    batch = [files[i % 3] for i in range(8)]
    fs = info.pixel_bytes
    for gpu in (False, True):
        for device in (False, True):
            out = torch.zeros(len(batch) * fs, dtype=torch.uint8, device="cuda") if device else np.zeros(len(batch) * fs, np.uint8)
            ctx.jpeg_decode_batch(batch, out, fs, threads=3, frames_per_chunk=3, gpu_entropy=gpu)
            got = out.cpu().numpy() if device else out
            for f in range(len(batch)):
                for i, plane in enumerate(info.planes(got[f * fs:(f + 1) * fs])):
                    assert np.array_equal(plane, wants[f % 3].plane(i)), (gpu, device, f, i)
        fs4 = 3 * info.width * info.height
        out4 = np.zeros(len(batch) * fs4, np.uint8)
        ctx.jpeg_decode_batch(batch, out4, fs4, threads=2, frames_per_chunk=3, yuv444=True, gpu_entropy=gpu)
        for f in range(len(batch)):
            y, u, v = wants[f % 3].get_yuv_frame()
            fr = out4[f * fs4:(f + 1) * fs4].reshape(3, info.height, info.width)
            assert np.array_equal(fr[0], y) and np.array_equal(fr[1], orc.supersample_hv2(u)) and \
                np.array_equal(fr[2], orc.supersample_hv2(v)), (gpu, f)


def test_the_wide_kernel_alone_is_unchanged_on_ordinary_blocks(ctx):
    """hvc_set_decode_kernel(ctx, 2): every block through k_decode_wide -- the 63-bit reading is the identity on everything
    16-bit coefficients and tables reach; checked on adversarial int16 records against the restatement"""
    rng = np.random.Generator(np.random.PCG64(5))
    bw, bh = 9, 5
    coefs = rng.integers(-32768, 32768, size=(bh, bw, 64)).astype(np.int16)
    q = rng.integers(1, 65536, size=64).astype(np.uint16)
    want = orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8)
    out = np.zeros((bh * 8, bw * 8), dtype=np.uint8)
    ctx.set_decode_kernel(2)
    try:
        ctx.dequant_idct_recon(coefs, q, bw, bh, 1, out)
    finally:
        ctx.set_decode_kernel(0)
    assert np.array_equal(out, want)
