"""GPU parity tests of the encode path (K3: level shift + forward DCT + quantise +
zig-zag) and the 4:2:0 -> 4:4:4 upsample (K2) through the C ABI, against the
CPU oracle.  Bit-exact."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def gpu_fdct_quant(ctx, plane, q, stride=None):
    h, w = plane.shape
    bw, bh = (stride and (w // 8) or w // 8), h // 8
    out = np.zeros((bh, bw, 64), dtype=np.int16)
    ctx.fdct_quant(np.ascontiguousarray(plane), q, bw, bh, 1, out)
    return out


def test_g1_chen_forward_kat(ctx):
    """test_chen_dct.ml:47-87 input block: q=1 turns quant_and_scale into the test's
    own '/4 rounded' (x>0 ? (x+2)/4 : (x-2)/4), in zig-zag order."""
    g = golden_json("g1_chen_dct.json")
    pix = (np.array(g["input"]) + 128).astype(np.uint8).reshape(8, 8)
    got = gpu_fdct_quant(ctx, pix, np.ones(64, dtype=np.uint16)).reshape(64)
    zi = orc.zigzag_inverse()
    assert [int(got[k]) for k in range(64)] == [g["fdct_div4_rounded"][zi[k]] for k in range(64)]


@pytest.mark.parametrize("fn,chroma", [("mini64x64.420", 420), ("mini64x64.422", 422), ("mini64x64.444", 444)])
@pytest.mark.parametrize("quality", [1, 30, 75, 95, 100])
def test_reference_frames_all_qualities(ctx, fn, chroma, quality):
    y, u, v = orc.split_yuv(golden_bytes(fn), 64, 64, chroma)
    _, coefs = orc.encode_yuv(y, u, v, 64, 64, chroma, quality, want_coefs=True)
    ql = orc.quant_scale(orc.quant_luma(), quality).astype(np.uint16)
    qc = orc.quant_scale(orc.quant_chroma(), quality).astype(np.uint16)
    for plane, q, want in ((y, ql, coefs[0]), (u, qc, coefs[1]), (v, qc, coefs[2])):
        assert np.array_equal(gpu_fdct_quant(ctx, plane, q), want)


@pytest.mark.parametrize("bw,bh", [(1, 1), (3, 5), (33, 9), (240, 135), (257, 2)])
def test_ragged_sizes_and_extreme_pixels(ctx, bw, bh):
    rng = np.random.Generator(np.random.PCG64(bw * 131 + bh))
    pix = synth_pixels(bw * 7 + bh, bh * 8, bw * 8)
    pix[:8, :8] = 255
    if bw > 1:
        pix[:8, 8:16] = 0
    if bh > 1:
        yy, xx = np.mgrid[0:8, 0:8]
        pix[8:16, :8] = np.where((yy + xx) % 2 == 0, 255, 0)  # checkerboard: largest AC energy
    q = rng.integers(1, 256, size=64).astype(np.uint16)
    want = orc.fdct_quant(pix, q, bw, bh).reshape(bh, bw, 64)
    assert np.array_equal(gpu_fdct_quant(ctx, pix, q), want)


@pytest.mark.parametrize("t", [1, 2, 3, 16, 255])
def test_forward_path_on_the_blocks_that_drive_it_hardest(ctx, t):
    """K3's c4 by v_mul_hi_i32_i24 and its one-fma quantiser (round 5) on the inputs that push every intermediate to its bound:
    all 64 two-level blocks whose signs follow one DCT basis function (each maximises one coefficient), both polarities, their
    products with a second basis function, single pixels, and dense random two-level blocks -- under constant tables of the
    smallest and largest divisors (quotients up to +-2^13 at t = 1) -- against the model restatement."""
    yy, xx = np.mgrid[0:8, 0:8]
    blocks = []
    for v in range(8):
        for u in range(8):
            basis = np.cos((2 * xx + 1) * u * np.pi / 16) * np.cos((2 * yy + 1) * v * np.pi / 16)
            for lo, hi in ((0, 255), (255, 0), (1, 254), (127, 128)):
                blocks.append(np.where(basis >= 0, hi, lo))
            blocks.append(np.where(basis * np.cos((2 * xx + 1) * ((u + 3) % 8) * np.pi / 16) >= 0, 255, 0))
    for k in range(64):
        b = np.zeros((8, 8), dtype=np.int64)
        b[k // 8, k % 8] = 255
        blocks += [b, 255 - b]
    rng = np.random.Generator(np.random.PCG64(1000 + t))
    blocks += [np.where(rng.random((8, 8)) < 0.5, 0, 255) for _ in range(192)]
    n = len(blocks)
    plane = np.concatenate([b.astype(np.uint8) for b in blocks], axis=1)   # one row of n blocks
    q = np.full(64, t, dtype=np.uint16)
    want = orc.fdct_quant(plane, q, n, 1)
    got = np.zeros(n * 64, dtype=np.int16)
    ctx.fdct_quant(np.ascontiguousarray(plane), q, n, 1, 1, got)
    assert np.array_equal(got, want.reshape(-1))
    assert np.abs(want).max() > 900 // t   # (the cases do reach the largest quotients a table allows: 1024 / t)


def test_frame_batch_encode_then_decode_roundtrip(ctx):
    """encode -> decode of a 4:2:0 frame batch on the GPU equals the oracle's
    encode -> decode (both directions exact => identical pixels)."""
    import video_coding_amd as hvc
    planes = [(6, 4, 0), (3, 2, 1), (3, 2, 1)]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    n_frames = 5
    ql = orc.quant_scale(orc.quant_luma(), 60).astype(np.uint16)
    qc = orc.quant_scale(orc.quant_chroma(), 60).astype(np.uint16)
    qtabs = np.stack([ql, qc])
    pixels = np.zeros(n_frames * pfs, dtype=np.uint8)
    for f in range(n_frames):
        for s in specs:
            n = s["blocks_w"] * s["blocks_h"] * 64
            pixels[f * pfs + s["plane_offset"]:f * pfs + s["plane_offset"] + n] = synth_pixels(
                f * 10 + s["blocks_w"], s["blocks_h"] * 8, s["blocks_w"] * 8).reshape(-1)
    coefs = np.zeros(n_frames * cfs, dtype=np.int16)
    ctx.encode_frames(pixels, pfs, qtabs, specs, n_frames, coefs, cfs)
    recon = np.zeros_like(pixels)
    ctx.decode_frames(coefs, cfs, qtabs, specs, n_frames, recon, pfs)
    for f in range(n_frames):
        for s in specs:
            bw, bh = s["blocks_w"], s["blocks_h"]
            n = bw * bh * 64
            src = pixels[f * pfs + s["plane_offset"]:f * pfs + s["plane_offset"] + n].reshape(bh * 8, bw * 8)
            wc = orc.fdct_quant(src, qtabs[s["qtab"]], bw, bh)
            assert np.array_equal(coefs[f * cfs + s["coef_offset"]:f * cfs + s["coef_offset"] + n], wc)
            wp = orc.dequant_idct_recon(wc, qtabs[s["qtab"]], bw, bh)
            assert np.array_equal(recon[f * pfs + s["plane_offset"]:f * pfs + s["plane_offset"] + n], wp)


@pytest.mark.parametrize("align", [4096, 65536, "auto"])
def test_planes_and_frames_on_aligned_boundaries_give_the_same_records(ctx, align):
    """the resident layout of bench.py (hvc.frame_layout(..., align): every plane of every frame -- pixels and coefficients -- on
    an `align`-byte boundary) through both block stages on device memory: the planes gathered tight equal the tight layout's,
    which equal the oracle's; the padding stays as it was"""
    import torch
    import video_coding_amd as hvc
    planes = [(15, 9, 0), (8, 5, 1), (8, 5, 1)]
    tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes, align=align)
    a = hvc.hvc.layout_alignment(planes) if align == "auto" else align
    assert all(s["plane_offset"] % a == 0 and (2 * s["coef_offset"]) % a == 0 for s in specs) and pfs % a == 0 and (2 * cfs) % a == 0
    n = 7
    qtabs = np.stack([orc.quant_scale(orc.quant_luma(), 60), orc.quant_scale(orc.quant_chroma(), 60)]).astype(np.uint16)
    tight = np.stack([np.concatenate([synth_pixels(f * 10 + bw, bh * 8, bw * 8).reshape(-1) for bw, bh, _ in planes]) for f in range(n)])
    d_tight = torch.from_numpy(tight).cuda()
    d_pix = hvc.hvc.spread_records(d_tight, tspecs, specs, pfs, "plane_offset")
    d_pix[d_pix == 0] = 0   # (no-op: the padding is zero)
    d_coefs = torch.full((n, cfs), 0x5a5a, dtype=torch.int16, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.encode_frames(d_pix, pfs, qtabs, hvc.hvc.components(specs), n, d_coefs, cfs)
        d_out = torch.full((n, pfs), 0xa5, dtype=torch.uint8, device="cuda")
        ctx.decode_frames(d_coefs, cfs, qtabs, hvc.hvc.components(specs), n, d_out, pfs)
        ctx.synchronize()
    finally:
        ctx.reset_stream()
    got_c = hvc.hvc.tight_records(d_coefs, specs, "coef_offset").cpu().numpy()
    got_p = hvc.hvc.tight_records(d_out, specs, "plane_offset").cpu().numpy()
    for f in range(n):
        off = 0
        for bw, bh, qt in planes:
            m = bw * bh * 64
            wc = orc.fdct_quant(tight[f, off:off + m].reshape(bh * 8, bw * 8), qtabs[qt], bw, bh).reshape(-1)
            assert np.array_equal(got_c[f, off:off + m], wc)
            assert np.array_equal(got_p[f, off:off + m], orc.dequant_idct_recon(wc, qtabs[qt], bw, bh).reshape(-1))
            off += m
    # what lies between the planes was not written: still the fill pattern
    mask = np.ones(pfs, dtype=bool)
    for s in specs:
        mask[s["plane_offset"]:s["plane_offset"] + s["blocks_w"] * s["blocks_h"] * 64] = False
    assert (d_out.cpu().numpy()[:, mask] == 0xa5).all()
    cmask = np.ones(cfs, dtype=bool)
    for s in specs:
        cmask[s["coef_offset"]:s["coef_offset"] + s["blocks_w"] * s["blocks_h"] * 64] = False
    assert (d_coefs.cpu().numpy()[:, cmask] == 0x5a5a).all()


def test_g7_upsample_kat(ctx):
    g = golden_json("g7_upsample.json")["cases"]["444<->420"]
    f420, back = g[1], g[2]
    for rows, want in ((f420[4:6], back[4:8]), (f420[6:8], back[8:12])):
        src = np.array(rows, dtype=np.uint8)
        dst = np.zeros((4, 4), dtype=np.uint8)
        ctx.upsample420(src, 2, 2, dst)
        assert dst.tolist() == want


@pytest.mark.parametrize("cw,ch", [(1, 1), (2, 3), (5, 7), (26, 22), (960, 540), (33, 2)])
def test_upsample_sizes(ctx, cw, ch):
    rng = np.random.Generator(np.random.PCG64(cw * 1000 + ch))
    n = 3
    src = rng.integers(0, 256, size=(n, ch, cw)).astype(np.uint8)
    dst = np.zeros((n, 2 * ch, 2 * cw), dtype=np.uint8)
    ctx.upsample420(src, cw, ch, dst, n_planes=n)
    for p in range(n):
        assert np.array_equal(dst[p], orc.supersample_hv2(src[p])), p


@pytest.mark.parametrize("cw,ch,sstride,dstride", [(8, 1, 8, 16), (16, 3, 24, 48), (120, 67, 128, 256), (960, 540, 960, 1920),
                                                   (968, 5, 976, 1936 + 16), (4, 4, 8, 16), (12, 2, 16, 32)])
def test_upsample_wide_variant_and_strides(ctx, cw, ch, sstride, dstride):
    """cw % 8 == 0 with 8 / 16-byte aligned rows takes k_upsample420_x8 (packed v_lerp_u8 arithmetic);
    padded strides and plane strides, device memory; cw % 8 != 0 stays on the 4-sample kernel."""
    import torch
    rng = np.random.Generator(np.random.PCG64(cw * 7 + ch))
    n = 4
    sps, dps = sstride * ch + 64, dstride * 2 * ch + 128
    src = rng.integers(0, 256, size=(n * sps,), dtype=np.uint8)
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.full((n * dps,), 0x5A, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.upsample420(d_src, cw, ch, d_dst, n_planes=n, src_stride=sstride, dst_stride=dstride, src_plane_stride=sps,
                    dst_plane_stride=dps)
    ctx.synchronize()
    got = d_dst.cpu().numpy()
    for p in range(n):
        plane = src[p * sps:p * sps + sstride * ch].reshape(ch, sstride)[:, :cw]
        out = got[p * dps:p * dps + dstride * 2 * ch].reshape(2 * ch, dstride)
        assert np.array_equal(out[:, :2 * cw], orc.supersample_hv2(np.ascontiguousarray(plane))), p
        assert (out[:, 2 * cw:] == 0x5A).all()   # row padding untouched
        assert (got[p * dps + dstride * 2 * ch:(p + 1) * dps] == 0x5A).all()


def test_encoder_every_constant_and_two_level_block(ctx):
    """Forward path on structured extremes: all 256 flat blocks, and every pair of levels (0 / v, v / 255)
    laid out as vertical, horizontal and checkerboard edges -- the inputs that drive each Chen rotation
    to its largest outputs -- at qualities 100 (divisor 4) and 50."""
    blocks = [np.full((8, 8), v, dtype=np.uint8) for v in range(256)]
    yy, xx = np.mgrid[0:8, 0:8]
    for v in range(0, 256, 5):
        for lo, hi in ((0, v), (v, 255)):
            for mask in (xx < 4, yy < 4, (xx + yy) % 2 == 0, xx % 2 == 0, yy == 0, (xx == 7) & (yy == 7)):
                blocks.append(np.where(mask, lo, hi).astype(np.uint8))
    n = len(blocks)
    plane = np.concatenate(blocks, axis=1)  # 8 x 8n
    for quality in (100, 50):
        q = orc.quant_scale(orc.quant_luma(), quality).astype(np.uint16)
        want = orc.fdct_quant(plane, q, n, 1)
        got = np.zeros(n * 64, dtype=np.int16)
        ctx.fdct_quant(np.ascontiguousarray(plane), q, n, 1, 1, got)
        assert np.array_equal(got, want.reshape(-1)), quality


def test_encoder_rejects_wide_tables(ctx):
    import video_coding_amd as hvc
    q = np.full(64, 256, dtype=np.uint16)
    with pytest.raises(hvc.HvcError) as e:
        ctx.fdct_quant(np.zeros((8, 8), dtype=np.uint8), q, 1, 1, 1, np.zeros(64, dtype=np.int16))
    assert e.value.code == -5


@pytest.mark.parametrize("quality", [1, 30, 75, 100])
@pytest.mark.parametrize("device", [False, True])
def test_encoder_reconstruction_error_path(ctx, quality, device):
    """a16: Encoder.encode_block with ~compute_reconstruction_error:true (encoder.ml:110-125, 195-205) --
    hvc_encode_frames_recon's coefficients, recon planes (max 0 (min 255 (idct + 128))) and per-sample error
    (abs (recon - input)) against the restated Encoder.dequant / idct / recon, on a 4:2:0 frame batch with
    padded strides (bytes outside the planes must stay untouched), incl. saturating content at quality 1."""
    import torch
    import video_coding_amd as hvc
    planes = [(6, 4, 0), (3, 2, 1), (3, 2, 1)]
    pad = 16
    specs, co, po = [], 0, 0
    for bw, bh, qt in planes:
        specs.append(dict(blocks_w=bw, blocks_h=bh, qtab=qt, coef_offset=co, plane_offset=po, stride=bw * 8 + pad))
        co += bw * bh * 64
        po += (bw * 8 + pad) * bh * 8
    cfs, pfs, n = co + 64, po + 40, 3
    qtabs = np.stack([orc.quant_scale(orc.quant_luma(), quality), orc.quant_scale(orc.quant_chroma(), quality)]).astype(np.uint16)
    pix = np.full(n * pfs, 0xEE, dtype=np.uint8)
    want = []
    for f in range(n):
        for i, (s, (bw, bh, qt)) in enumerate(zip(specs, planes)):
            p = synth_pixels(40 * f + i, bh * 8, bw * 8)
            if f == 2:  # hard edges: the reconstruction overshoots 0 / 255 and the clamp of Encoder.recon acts
                p = np.where(p > 127, 255, 0).astype(np.uint8)
            v = pix[f * pfs + s["plane_offset"]:][:(bw * 8 + pad) * bh * 8].reshape(bh * 8, bw * 8 + pad)
            v[:, :bw * 8] = p
            want.append(orc.encode_recon(p, qtabs[qt], bw, bh))
    coefs = np.zeros(n * cfs, dtype=np.int16)
    recon = np.full(n * pfs, 0x11, dtype=np.uint8)
    err = np.full(n * pfs, 0x22, dtype=np.uint8)
    if device:
        d = [torch.from_numpy(a).cuda() for a in (pix, coefs, recon, err)]
        torch.cuda.synchronize()
        ctx.encode_frames_recon(d[0], pfs, qtabs, specs, n, d[1], cfs, d[2], d[3])
        ctx.synchronize()
        coefs, recon, err = (t.cpu().numpy() for t in d[1:])
        # the error plane alone: the reconstruction goes to scratch
        e2 = torch.full((n * pfs,), 0x22, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.encode_frames_recon(d[0], pfs, qtabs, specs, n, d[1], cfs, None, e2)
        ctx.synchronize()
        assert np.array_equal(e2.cpu().numpy(), err)
    else:
        ctx.encode_frames_recon(pix, pfs, qtabs, specs, n, coefs, cfs, recon, err)
    k, clamped = 0, False
    for f in range(n):
        for s, (bw, bh, qt) in zip(specs, planes):
            wc, wr, we = want[k]
            k += 1
            assert np.array_equal(coefs[f * cfs + s["coef_offset"]:][:bw * bh * 64], wc), (f, s)
            for got, w, fill in ((recon, wr, 0x11), (err, we, 0x22)):
                v = got[f * pfs + s["plane_offset"]:][:(bw * 8 + pad) * bh * 8].reshape(bh * 8, bw * 8 + pad)
                assert np.array_equal(v[:, :bw * 8], w.reshape(bh * 8, bw * 8)), (f, s)
                assert (v[:, bw * 8:] == fill).all()
            clamped |= bool((wr == 0).any() and (wr == 255).any())
    assert clamped
    # the reconstruction is what the decoder makes of these coefficients
    f0 = specs[0]
    got = recon[f0["plane_offset"]:][:(6 * 8 + pad) * 32].reshape(32, 6 * 8 + pad)[:, :48]
    assert np.array_equal(got, orc.dequant_idct_recon(coefs[:6 * 4 * 64], qtabs[0], 6, 4).reshape(32, 48))
