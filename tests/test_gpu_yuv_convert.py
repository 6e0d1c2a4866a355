"""The rest of `oyuv convert` on the GPU (csrc/hvc_yuv.hip): Planar_444.subsample_hv2 / subsample_h2 / supersample_h2, Yuv.crop,
Packed_422 and Oconv.main's per-frame pipeline through the C ABI, against the restated tools (oracle/) and the reference's
own expect-test frames (G7)."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


A = lambda rows: np.ascontiguousarray(np.array(rows, dtype=np.uint8))


def test_g7_kats_on_the_gpu(ctx):
    """tools/src/planar_444.ml:139-249: the 4 x 4 frame through 4:4:4 -> 4:2:0 -> 4:4:4 and 4:4:4 -> 4:2:2 -> 4:4:4"""
    g = golden_json("g7_upsample.json")["cases"]
    f444, f420, back = g["444<->420"]
    for lo, hi, c0, c1 in ((4, 8, 4, 6), (8, 12, 6, 8)):
        out = np.zeros((2, 2), np.uint8)
        ctx.subsample420(A(f444[lo:hi]), 4, 4, out)
        assert out.tolist() == f420[c0:c1]
        up = np.zeros((4, 4), np.uint8)
        ctx.upsample420(A(f420[c0:c1]), 2, 2, up)
        assert up.tolist() == back[lo:hi]
    f444, f422, back = g["444<->422"]
    for lo, hi in ((4, 8), (8, 12)):
        out = np.zeros((4, 2), np.uint8)
        ctx.subsample422(A(f444[lo:hi]), 4, 4, out)
        assert out.tolist() == f422[lo:hi]
        up = np.zeros((4, 4), np.uint8)
        ctx.upsample422(A(f422[lo:hi]), 2, 4, up)
        assert up.tolist() == back[lo:hi]


def test_g7_packed_kat_on_the_gpu(ctx):
    """tools/src/packed_422.ml:56-104 through hvc_yuv_convert.  Oconv always passes through a 4:4:4 frame (oconv.ml:12-51), so
    the chroma samples of a 4:2:2 -> YUY2 conversion are supersample_h2 then subsample_h2 of the frame's (not the identity:
    avg2 50 55 = 53); the luma samples and every byte's PLACE are the expect test's."""
    import video_coding_amd as hvc
    g = golden_json("g7_packed422.json")
    y, u, v = A(g["frame"][0:4]), A(g["frame"][4:8]), A(g["frame"][8:12])
    frame = np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)])
    packed = np.zeros(32, np.uint8)
    ctx.yuv_convert(frame, 422, (4, 4), packed, hvc.hvc.YUV_FORMATS["YUY2"], (4, 4))
    u2, v2 = (orc.subsample_h2(orc.supersample_h2(p), 2, 4) for p in (u, v))    # (pinned by G7's 444<->422 frames)
    kat = np.array(g["packed"], dtype=np.uint8).reshape(4, 2, 4)
    got = packed.reshape(4, 2, 4)
    assert np.array_equal(got[:, :, 0], kat[:, :, 0]) and np.array_equal(got[:, :, 2], kat[:, :, 2])   # Y0, Y1 as in the expect test
    assert np.array_equal(got[:, :, 1], u2) and np.array_equal(got[:, :, 3], v2)                       # U and V at their places
    assert packed.tobytes() == orc.oconv_frame(frame, 422, (4, 4), "YUY2", (4, 4))
    back = np.zeros(32, np.uint8)
    ctx.yuv_convert(np.array(g["packed"], dtype=np.uint8).reshape(-1), hvc.hvc.YUV_FORMATS["YUY2"], (4, 4), back, 422, (4, 4))
    assert np.array_equal(back[:16], y.reshape(-1))                                                     # the unpacked luma plane
    assert back.tobytes() == orc.oconv_frame(np.array(g["packed"], dtype=np.uint8).reshape(-1), "YUY2", (4, 4), 422, (4, 4))


@pytest.mark.parametrize("device", [False, True])
def test_plane_operations_on_random_planes(ctx, device):
    """every size class of the kernels: whole 8-sample groups and partial ones, odd sizes (the unused last column / row),
    one row, one column pair, padded strides, several planes -- host buffers and device buffers"""
    import torch
    rng = np.random.Generator(np.random.PCG64(11))

    def run(fn, src, dst_shape, *args, **kw):
        out = np.full(dst_shape, 0xAA, np.uint8)
        if device:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            d_src, d_out = torch.from_numpy(src.copy()).cuda(), torch.from_numpy(out.copy()).cuda()
            fn(d_src, *args, d_out, **kw)
            ctx.synchronize()
            ctx.reset_stream()
            return d_out.cpu().numpy()
        fn(src, *args, out, **kw)
        return out

    for (w, h) in ((2, 2), (3, 5), (16, 2), (18, 7), (64, 64), (250, 33), (256, 16), (1026, 9)):
        for n_planes, pad in ((1, 0), (3, 24)):
            src = rng.integers(0, 256, size=(n_planes, h, w + pad), dtype=np.uint8)
            planes = [src[p, :, :w] for p in range(n_planes)]
            kw = dict(n_planes=n_planes, src_stride=w + pad, src_plane_stride=h * (w + pad))
            # subsample_hv2
            if w >= 2 and h >= 2:
                dw, dh = w // 2, h // 2
                got = run(ctx.subsample420, src, (n_planes, dh, dw + 8), w, h, dst_stride=dw + 8, dst_plane_stride=dh * (dw + 8), **kw)
                for p in range(n_planes):
                    assert np.array_equal(got[p, :, :dw], orc.subsample_hv2(planes[p], dw, dh)), (w, h, p)
                    assert (got[p, :, dw:] == 0xAA).all()          # the caller's padding is not touched
            # subsample_h2
            dw = w // 2
            got = run(ctx.subsample422, src, (n_planes, h, dw + 8), w, h, dst_stride=dw + 8, dst_plane_stride=h * (dw + 8), **kw)
            for p in range(n_planes):
                assert np.array_equal(got[p, :, :dw], orc.subsample_h2(planes[p], dw, h)), (w, h, p)
                assert (got[p, :, dw:] == 0xAA).all()
            # supersample_h2
            got = run(ctx.upsample422, src, (n_planes, h, 2 * w + 16), w, h, dst_stride=2 * w + 16, dst_plane_stride=h * (2 * w + 16), **kw)
            for p in range(n_planes):
                assert np.array_equal(got[p, :, :2 * w], orc.supersample_h2(planes[p])), (w, h, p)
                assert (got[p, :, 2 * w:] == 0xAA).all()
            # Yuv.crop: inside, shifted out of every edge, larger than the source
            for (dw, dh, x, y) in ((w, h, 0, 0), (max(1, w - 3), max(1, h - 1), 2, 1), (w + 5, h + 4, -3, -2), (9, 3, w - 2, h - 1)):
                got = run(lambda s_, dw_, dh_, o_, **k: ctx.crop_planes(s_, w, h, x, y, o_, dw_, dh_, **k), src, (n_planes, dh, dw + 8),
                          dw, dh, dst_stride=dw + 8, dst_plane_stride=dh * (dw + 8), **kw)
                for p in range(n_planes):
                    assert np.array_equal(got[p, :, :dw], orc.crop_plane(planes[p], dw, dh, x, y)), (w, h, dw, dh, x, y, p)
                    assert (got[p, :, dw:] == 0xAA).all()


FORMATS = [420, 422, 444, "YUY2", "UYVY", "YVYU"]


@pytest.mark.parametrize("fmt_in", FORMATS)
@pytest.mark.parametrize("fmt_out", FORMATS)
def test_oconv_pipeline_every_format_pair(ctx, fmt_in, fmt_out):
    """Oconv.main's loop body: same size, a crop at an offset, a larger frame (edge replication) -- two frames per call"""
    import video_coding_amd as hvc
    F = lambda f: f if isinstance(f, int) else hvc.hvc.YUV_FORMATS[f]
    rng = np.random.Generator(np.random.PCG64(FORMATS.index(fmt_in) * 7 + FORMATS.index(fmt_out)))
    big = ((420, 444), (444, 420), (422, 444), (444, 422), ("YUY2", 420), (420, "UYVY"), ("YVYU", "YUY2"), (420, 420), (422, "YVYU"))
    for (size_in, size_out, off) in (((64, 48), (64, 48), (0, 0)), ((64, 48), (52, 44), (0, 0)), ((70, 34), (32, 16), (9, 5)),
                                     ((32, 16), (48, 40), (-6, -4)), ((96, 40), (48, 24), (16, 6)), ((96, 40), (60, 30), (32, 3)),
                                     ((100, 40), (48, 24), (16, 6)), ((1920, 1080), (1920, 1080), (0, 0)),
                                     ((1920, 1080), (1280, 720), (320, 180)), ((1920, 1080), (1272, 718), (321, 181))):
        # (the crop windows at offsets of 16 and 32 columns of a 96-column source are read in place by the sub-sampling
        # kernels; 100 columns or an odd offset take the materialised crop)
        if size_in[0] > 1000 and (fmt_in, fmt_out) not in big:
            continue
        n_in = hvc.hvc.yuv_frame_bytes(F(fmt_in), *size_in)
        n_out = hvc.hvc.yuv_frame_bytes(F(fmt_out), *size_out)
        frames = rng.integers(0, 256, size=(2, n_in), dtype=np.uint8)
        out = np.zeros((2, n_out), np.uint8)
        ctx.yuv_convert(frames, F(fmt_in), size_in, out, F(fmt_out), size_out, offset=off, n_frames=2)
        for f in range(2):
            want = np.frombuffer(orc.oconv_frame(frames[f], fmt_in, size_in, fmt_out, size_out, off), dtype=np.uint8)
            assert want.size == n_out and np.array_equal(out[f], want), (fmt_in, fmt_out, size_in, size_out, off, f)


def test_sizes_the_tools_raise_on(ctx):
    """Yuv.assert_is_420 / _422 (tools/src/yuv.ml:90-116): an odd width with a subsampled format, an odd height with 4:2:0"""
    import video_coding_amd as hvc
    buf, out = np.zeros(1 << 16, np.uint8), np.zeros(1 << 16, np.uint8)
    for (fi, si, fo, so) in ((420, (63, 48), 444, (63, 48)), (420, (64, 47), 444, (64, 47)), (444, (64, 48), 422, (51, 40)),
                             (444, (64, 48), 420, (50, 41)), (1, (33, 8), 444, (32, 8))):
        with pytest.raises(hvc.HvcError) as e:
            ctx.yuv_convert(buf, fi, si, out, fo, so)
        assert e.value.code == -1
        with pytest.raises(ValueError):
            orc.oconv_frame(buf, {1: "YUY2"}.get(fi, fi), si, {1: "YUY2"}.get(fo, fo), so)
    ctx.yuv_convert(buf, 444, (63, 47), out, 444, (31, 15))     # 4:4:4 takes any size


def test_device_resident_frames(ctx):
    import torch
    rng = np.random.Generator(np.random.PCG64(3))
    w, h, n = 1920, 1080, 4
    frames = rng.integers(0, 256, size=(n, w * h * 3), dtype=np.uint8)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros((n, w * h * 3 // 2), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.yuv_convert(d_in, 444, (w, h), d_out, 420, (w, h), n_frames=n)
    ctx.synchronize()
    ctx.reset_stream()
    got = d_out.cpu().numpy()
    for f in range(n):
        assert np.array_equal(got[f], np.frombuffer(orc.oconv_frame(frames[f], 444, (w, h), 420, (w, h)), dtype=np.uint8)), f
