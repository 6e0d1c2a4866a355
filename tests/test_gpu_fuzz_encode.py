"""Randomised sweep of the whole-frame encoder and of the fused 4:4:4 output against the model restatement: frame sizes
1..300 (odd ones, sizes the model's encoder refuses included), the three samplings, qualities 1..100, pixel content from
flat to full-range noise -- hvc_jpeg_encode byte-identical to Encoder.encode_4xx or refused where the model raises, the
file decoded back to the model's planes, hvc_jpeg_decode_yuv444 equal to decode_a_frame + Planar_444.of_420."""
import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def content(rng, h, w):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return np.full((h, w), int(rng.integers(0, 256)), dtype=np.uint8)
    if kind == 1:
        return rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    if kind == 2:   # extremes only: the largest coefficients a frame can give
        return (rng.integers(0, 2, size=(h, w)) * 255).astype(np.uint8)
    if kind == 3:   # smooth ramp with a little noise
        yy, xx = np.mgrid[0:h, 0:w]
        return np.clip((xx * 3 + yy * 2) % 256 + rng.integers(-3, 4, size=(h, w)), 0, 255).astype(np.uint8)
    blocks = rng.integers(0, 256, size=((h + 7) // 8, (w + 7) // 8), dtype=np.uint8)
    return np.kron(blocks, np.ones((8, 8), dtype=np.uint8))[:h, :w]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_frames_encode_like_the_model(ctx, seed):
    import video_coding_amd as m
    rng = np.random.Generator(np.random.PCG64(seed))
    encoded = refused = 0
    for it in range(80):
        w, h = int(rng.integers(1, 301)), int(rng.integers(1, 301))
        chroma = int(rng.choice([420, 422, 444]))
        q = int(rng.integers(1, 101))
        cw, ch = orc.chroma_dims(chroma, w, h)
        if cw < 1 or ch < 1:
            continue
        y, u, v = content(rng, h, w), content(rng, ch, cw), content(rng, ch, cw)
        try:
            want = orc.encode_yuv(y, u, v, w, h, chroma, q)
        except Exception:
            want = None   # "[Plane.get] out of bounds": widths / heights of 16 k + 1 with subsampling (encoder.ml:476-505)
        try:
            got = ctx.jpeg_encode(y, u, v, w, h, chroma, q)
        except m.HvcError:
            got = None
        assert (got is None) == (want is None), (it, w, h, chroma, q)
        if got is None:
            refused += 1
            continue
        assert got == want, (it, w, h, chroma, q)
        encoded += 1
        info, pixels = ctx.jpeg_decode(got)
        d = orc.Decoder(got)
        d.decode()
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, d.plane(i)), (it, i)
        if chroma == 420 and w % 2 == 0 and h % 2 == 0:   # (the fused output takes even sizes: a chroma sample covers 2 x 2)
            frame = ctx.jpeg_decode_yuv444(got)[1]
            yy, uu, vv = d.get_yuv_frame()
            want444 = np.concatenate([yy.reshape(-1), orc.supersample_hv2(uu)[:h, :w].reshape(-1), orc.supersample_hv2(vv)[:h, :w].reshape(-1)])
            assert np.array_equal(np.asarray(frame).reshape(-1), want444), (it, w, h)
    assert encoded > 40, (encoded, refused)
