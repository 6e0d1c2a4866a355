"""Randomised frame layouts through the record-level entry points against the model restatement: 1..4 components of random
block dimensions, planes with padded strides and gaps between them, records with padding between frames, 1..4 quantiser
tables (8- and 16-bit entries, zeros on the decode side), several frames per call, host and device memory; coefficients
from the forward path, sparse, dense and extreme (the int64 fix-up).  hvc_decode_frames, hvc_encode_frames, hvc_upsample420."""
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


def random_layout(rng):
    n_comp = int(rng.integers(1, 5))
    n_q = int(rng.integers(1, 5))
    specs, co, po = [], 0, 0
    for _ in range(n_comp):
        bw, bh = int(rng.integers(1, 41)), int(rng.integers(1, 25))
        stride = bw * 8 + 8 * int(rng.integers(0, 4))            # padded rows (multiples of 8: the ABI's alignment rule)
        co += 8 * int(rng.integers(0, 9))                        # gaps between the component records
        po += 8 * int(rng.integers(0, 9))
        specs.append(dict(blocks_w=bw, blocks_h=bh, qtab=int(rng.integers(0, n_q)), coef_offset=co, plane_offset=po, stride=stride))
        co += bw * bh * 64
        po += stride * bh * 8
    return specs, co + 8 * int(rng.integers(0, 5)), po + 8 * int(rng.integers(0, 5)), n_q


def random_tables(rng, n_q, zeros):
    q = np.zeros((n_q, 64), dtype=np.uint16)
    for t in range(n_q):
        kind = int(rng.integers(0, 3))
        q[t] = rng.integers(1, 256, size=64) if kind == 0 else rng.integers(1, 17, size=64) if kind == 1 else rng.integers(1, 4000, size=64)
        if zeros and rng.integers(0, 3) == 0:
            q[t, rng.integers(0, 64, size=3)] = 0
    return q


@pytest.mark.parametrize("seed", [1, 2])
def test_decode_frames_over_random_layouts(ctx, seed):
    import torch
    rng = np.random.Generator(np.random.PCG64(seed))
    for it in range(40):
        specs, cfs, pfs, n_q = random_layout(rng)
        q = random_tables(rng, n_q, zeros=True)
        n_frames = int(rng.integers(1, 5))
        coefs = np.zeros(n_frames * cfs, dtype=np.int16)
        for f in range(n_frames):
            for s in specs:
                n = s["blocks_w"] * s["blocks_h"] * 64
                kind = int(rng.integers(0, 4))
                if kind == 0:     # sparse, small
                    c = (rng.random(n) < 0.1) * rng.integers(-30, 31, size=n)
                elif kind == 1:   # dense, moderate
                    c = rng.integers(-200, 201, size=n)
                elif kind == 2:   # extreme: whatever int16 holds
                    c = rng.choice(np.array([-32768, -2048, -1, 0, 1, 2047, 32767]), size=n)
                else:             # DC only
                    c = np.zeros(n, dtype=np.int64)
                    c[::64] = rng.integers(-1024, 1024, size=n // 64)
                coefs[f * cfs + s["coef_offset"]:f * cfs + s["coef_offset"] + n] = c.astype(np.int16)
        device = bool(rng.integers(0, 2))
        canary = 0xA5
        if device:
            d_c = torch.from_numpy(coefs).cuda()
            d_p = torch.full((n_frames * pfs,), canary, dtype=torch.uint8, device="cuda")
            ctx.decode_frames(d_c, cfs, q, specs, n_frames, d_p, pfs)
            torch.cuda.synchronize()
            pixels = d_p.cpu().numpy()
        else:
            pixels = np.full(n_frames * pfs, canary, dtype=np.uint8)
            ctx.decode_frames(coefs, cfs, q, specs, n_frames, pixels, pfs)
        touched = np.zeros(n_frames * pfs, dtype=bool)
        for f in range(n_frames):
            for s in specs:
                bw, bh, st = s["blocks_w"], s["blocks_h"], s["stride"]
                n = bw * bh * 64
                want = orc.dequant_idct_recon(coefs[f * cfs + s["coef_offset"]:][:n], q[s["qtab"]], bw, bh).reshape(bh * 8, bw * 8)
                base = f * pfs + s["plane_offset"]
                rows = base + np.arange(bh * 8)[:, None] * st + np.arange(bw * 8)[None, :]
                assert np.array_equal(pixels[rows], want), (it, f, s, device)
                touched[rows] = True
        assert (pixels[~touched] == canary).all(), (it, "padding written")


@pytest.mark.parametrize("seed", [3])
def test_encode_frames_over_random_layouts(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    for it in range(40):
        specs, cfs, pfs, n_q = random_layout(rng)
        q = random_tables(rng, n_q, zeros=False)
        q = np.minimum(q, 255).astype(np.uint16)   # (the encoder's tables are 8-bit: Quant_tables.scale clips to 255)
        n_frames = int(rng.integers(1, 4))
        pixels = rng.integers(0, 256, size=n_frames * pfs, dtype=np.uint8)
        if rng.integers(0, 2):
            pixels[:] = (pixels > 127) * 255         # extremes
        coefs = np.full(n_frames * cfs, 0x5A5A, dtype=np.int16)
        ctx.encode_frames(pixels, pfs, q, specs, n_frames, coefs, cfs)
        touched = np.zeros(n_frames * cfs, dtype=bool)
        for f in range(n_frames):
            for s in specs:
                bw, bh, st = s["blocks_w"], s["blocks_h"], s["stride"]
                base = f * pfs + s["plane_offset"]
                rows = base + np.arange(bh * 8)[:, None] * st + np.arange(bw * 8)[None, :]
                want = orc.fdct_quant(np.ascontiguousarray(pixels[rows]), q[s["qtab"]], bw, bh)
                at = f * cfs + s["coef_offset"]
                assert np.array_equal(coefs[at:at + want.size], want), (it, f, s)
                touched[at:at + want.size] = True
        assert (coefs[~touched] == 0x5A5A).all(), (it, "gaps written")


def test_upsample_over_random_sizes(ctx):
    rng = np.random.Generator(np.random.PCG64(5))
    for it in range(60):
        cw, ch = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        n = int(rng.integers(1, 4))
        src = rng.integers(0, 256, size=(n, ch, cw), dtype=np.uint8)
        dst = np.zeros((n, 2 * ch, 2 * cw), dtype=np.uint8)
        ctx.upsample420(src, cw, ch, dst, n_planes=n, src_plane_stride=cw * ch, dst_plane_stride=4 * cw * ch)
        for p in range(n):
            assert np.array_equal(dst[p], orc.supersample_hv2(src[p])), (it, p, cw, ch)


def test_fused_444_over_random_even_sizes(ctx):
    """hvc_decode_frames_yuv444 (block stage + crop + chroma upsample in one pass) at random even sizes -- tile seams, crops
    inside block rows, byte-path widths -- random frame counts, padded frame strides, a share of blocks outside the proven
    range (int64 fix-up + re-interpolation), host and device buffers, against decode -> crop -> supersample_hv2."""
    from test_gpu_yuv444 import expected444, geometry420, make_record, run, tables
    rng = np.random.Generator(np.random.PCG64(9))
    for it in range(30):
        width, height = 2 * int(rng.integers(1, 350)), 2 * int(rng.integers(1, 120))
        planes, qt = geometry420(width, height), tables(int(rng.choice([10, 50, 75, 95])))
        n = int(rng.integers(1, 4))
        adversarial = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
        recs = [make_record(100 * it + f, planes, qt, adversarial) for f in range(n)]
        fs = 3 * width * height + int(rng.integers(0, 3)) * 16
        got = run(ctx, recs, planes, qt, width, height, bool(rng.integers(0, 2)), frame_stride=fs)
        for f, rec in enumerate(recs):
            want = expected444(rec, planes, qt, width, height)
            assert np.array_equal(got[f][:want.size], want), (it, f, width, height, adversarial)
            assert (got[f][want.size:] == 0xA5).all(), (it, "padding written")
