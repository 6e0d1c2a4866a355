"""GPU parity tests of the decode path (K1) through the C ABI, against the CPU
oracle on the same inputs.  Bit-exact (integer work)."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import coef_planes_from_jpeg, synth_coefs
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def gpu_decode_plane(ctx, coefs, qtab, stride=None, fill=0):
    bh, bw = coefs.shape[:2]
    stride = stride or bw * 8
    out = np.full((bh * 8, stride), fill, dtype=np.uint8)
    ctx.dequant_idct_recon(np.ascontiguousarray(coefs), qtab, bw, bh, 1, out, stride=stride)
    return out


def test_g2_mouse480_blocks(ctx):
    """The six golden blocks of Mouse480 (test_decoder_accelerator.ml:209-376)."""
    g = golden_json("g2_mouse480_blocks.json")
    hdr = golden_json("mouse480_header.json")
    tabs = {t["table_identifier"]: np.array(t["elements"], dtype=np.uint16) for t in hdr["quant_tables"]}
    tq = {c[0]: c[3] for c in hdr["components_id_h_v_tq"]}
    s12 = lambda v: np.where(np.array(v) >= 2048, np.array(v) - 4096, np.array(v))
    prev = {}
    for blk in g["blocks"]:
        c = s12(blk["coefs_lo12"]).astype(np.int16)
        c[0] = blk["dc_pred_after"]  # absolute DC
        out = gpu_decode_plane(ctx, c.reshape(1, 1, 64), tabs[tq[blk["identifier"]]])
        assert out.reshape(64).tolist() == blk["recon"], blk["block_number"]


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
def test_real_jpeg_planes(ctx, fn):
    comps, _ = coef_planes_from_jpeg(golden_bytes(fn))
    for i, c in enumerate(comps):
        out = gpu_decode_plane(ctx, c["coefs"], c["qtab"])
        assert np.array_equal(out, c["plane"]), (fn, i)
    assert ctx.last_wide_blocks() == 0


def test_frame_batch_api_420(ctx):
    """hvc_decode_frames: 3 components, 2 tables, several frames in one launch."""
    import video_coding_amd as hvc
    comps, _ = coef_planes_from_jpeg(golden_bytes("Mouse480.jpg"))
    specs, cfs, pfs = hvc.hvc.frame_layout([(c["coefs"].shape[1], c["coefs"].shape[0], min(i, 1))
                                            for i, c in enumerate(comps)])
    n_frames = 3
    coefs = np.zeros(n_frames * cfs, dtype=np.int16)
    for f in range(n_frames):
        for s, c in zip(specs, comps):
            blk = c["coefs"] if f != 1 else c["coefs"][::-1, ::-1]  # frame 1: blocks permuted
            coefs[f * cfs + s["coef_offset"]: f * cfs + s["coef_offset"] + blk.size] = blk.reshape(-1)
    qtabs = np.stack([comps[0]["qtab"], comps[1]["qtab"]])
    pixels = np.zeros(n_frames * pfs, dtype=np.uint8)
    ctx.decode_frames(coefs, cfs, qtabs, specs, n_frames, pixels, pfs)
    for f in range(n_frames):
        for s, c in zip(specs, comps):
            bw, bh = s["blocks_w"], s["blocks_h"]
            got = pixels[f * pfs + s["plane_offset"]: f * pfs + s["plane_offset"] + bw * bh * 64].reshape(bh * 8, bw * 8)
            if f != 1:
                want = c["plane"]
            else:
                want = orc.dequant_idct_recon(c["coefs"][::-1, ::-1], c["qtab"], bw, bh).reshape(bh * 8, bw * 8)
            assert np.array_equal(got, want), (f, s)


@pytest.mark.parametrize("bw,bh", [(1, 1), (1, 7), (3, 5), (17, 31), (240, 136), (255, 3)])
def test_ragged_sizes_valid_data(ctx, bw, bh):
    q = orc.quant_scale(orc.quant_luma(), 75).astype(np.uint16)
    coefs, _ = synth_coefs(1234 + bw * 1000 + bh, bh, bw, q)
    want = orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8)
    got = gpu_decode_plane(ctx, coefs, q)
    assert np.array_equal(got, want)
    assert ctx.last_wide_blocks() == 0  # encoder-producible data never needs the 64-bit kernel


def test_padded_stride_keeps_padding(ctx):
    q = orc.quant_scale(orc.quant_chroma(), 50).astype(np.uint16)
    bw, bh = 5, 4
    coefs, _ = synth_coefs(99, bh, bw, q)
    stride = bw * 8 + 24
    got = gpu_decode_plane(ctx, coefs, q, stride=stride, fill=0xA5)
    want = orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8)
    assert np.array_equal(got[:, :bw * 8], want)
    assert (got[:, bw * 8:] == 0xA5).all()


@pytest.mark.parametrize("seed,qmax,amp", [(1, 255, 32767), (2, 255, 2047), (3, 16, 32767), (4, 1, 32767),
                                           (5, 255, 300), (6, 100, 1200)])
def test_adversarial_coefficients_take_wide_path_and_stay_exact(ctx, seed, qmax, amp):
    """Arbitrary int16 coefficients (far outside what an encoder can produce):
    the int32 kernel's guard must route what it cannot prove safe to the int64
    kernel; output equals the int64 oracle everywhere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bw, bh = 37, 9
    coefs = rng.integers(-amp, amp + 1, size=(bh, bw, 64)).astype(np.int16)
    coefs[0, 0, :] = 32767
    coefs[0, 1, :] = -32768
    coefs[0, 2, :] = 0
    coefs[0, 2, 0] = -32768
    q = rng.integers(1, qmax + 1, size=64).astype(np.uint16)
    q[0] = qmax
    want = orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8)
    got = gpu_decode_plane(ctx, coefs, q)
    assert np.array_equal(got, want)


def test_guard_boundary_sweep(ctx):
    """Scale one dense block up until the guard trips: exact on both sides."""
    rng = np.random.Generator(np.random.PCG64(7))
    base = rng.integers(-1, 2, size=64)
    q = np.ones(64, dtype=np.uint16)
    n = 400
    coefs = np.zeros((1, n, 64), dtype=np.int16)
    for i in range(n):
        coefs[0, i] = np.clip(base * (i * 8), -32768, 32767)
    want = orc.dequant_idct_recon(coefs, q, n, 1).reshape(8, n * 8)
    got = gpu_decode_plane(ctx, coefs, q)
    assert np.array_equal(got, want)
    assert 0 < ctx.last_wide_blocks() < n


@pytest.mark.parametrize("quality", [None, 75, 1])
def test_every_single_coefficient_value(ctx, quality):
    """Exhaustive over one-coefficient blocks: every zig-zag position x every value a baseline JPEG
    can code (-2048 .. 2047: the basis functions at every amplitude and sign), with a flat table
    of ones, Quant_tables.scale 75 and the coarsest table (quality 1: entries up to 255, where the
    largest values leave the int32 kernel's proven range and must take the int64 path)."""
    vals = np.arange(-2048, 2048, dtype=np.int16)
    coefs = np.zeros((64, vals.size, 64), dtype=np.int16)
    for k in range(64):
        coefs[k, :, k] = vals
    q = np.ones(64, dtype=np.uint16) if quality is None else orc.quant_scale(orc.quant_luma(), quality).astype(np.uint16)
    want = orc.dequant_idct_recon(coefs, q, vals.size, 64).reshape(64 * 8, vals.size * 8)
    got = gpu_decode_plane(ctx, coefs, q)
    assert np.array_equal(got, want)
    if quality == 1:
        assert ctx.last_wide_blocks() > 0


def test_dc_plus_one_ac_extremes(ctx):
    """DC at its extremes combined with every single AC position at +-1023 (the largest AC magnitude
    baseline Huffman coding can express) and the q = 50 table."""
    q = orc.quant_scale(orc.quant_luma(), 50).astype(np.uint16)
    blocks = []
    for dc in (-2048, -1, 0, 2047):
        for k in range(1, 64):
            for ac in (-1023, 1023):
                b = np.zeros(64, dtype=np.int16)
                b[0], b[k] = dc, ac
                blocks.append(b)
    coefs = np.stack(blocks).reshape(1, len(blocks), 64)
    want = orc.dequant_idct_recon(coefs, q, len(blocks), 1).reshape(8, len(blocks) * 8)
    assert np.array_equal(gpu_decode_plane(ctx, coefs, q), want)


@pytest.mark.parametrize("bw,bh", [(1, 1), (15, 1), (17, 3), (255, 3), (240, 136)])
def test_quarter_wavefront_kernel(ctx, bw, bh):
    """hvc_set_decode_kernel(3): the north star's mapping (one block per 16 lanes, LDS between the
    passes) gives the same bytes -- valid data without the fix-up path, adversarial data through it."""
    q = orc.quant_scale(orc.quant_chroma(), 60).astype(np.uint16)
    coefs, _ = synth_coefs(bw * 31 + bh, bh, bw, q)
    rng = np.random.Generator(np.random.PCG64(bw))
    adv = rng.integers(-2047, 2048, size=coefs.shape).astype(np.int16)
    ctx.set_decode_kernel(3)
    try:
        got = gpu_decode_plane(ctx, coefs, q)
        wide = ctx.last_wide_blocks()
        got_adv = gpu_decode_plane(ctx, adv, q)
        wide_adv = ctx.last_wide_blocks()
    finally:
        ctx.set_decode_kernel(0)
    assert np.array_equal(got, orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8))
    assert wide == 0
    assert np.array_equal(got_adv, orc.dequant_idct_recon(adv, q, bw, bh).reshape(bh * 8, bw * 8))
    assert wide_adv > 0


def test_sixteen_bit_quant_table_goes_wide(ctx):
    rng = np.random.Generator(np.random.PCG64(8))
    coefs = rng.integers(-50, 51, size=(2, 3, 64)).astype(np.int16)
    q = rng.integers(1, 65536, size=64).astype(np.uint16)
    want = orc.dequant_idct_recon(coefs, q, 3, 2).reshape(16, 24)
    assert np.array_equal(gpu_decode_plane(ctx, coefs, q), want)


def test_device_memory_path_torch(ctx):
    import torch
    q = orc.quant_scale(orc.quant_luma(), 75).astype(np.uint16)
    bw, bh, n = 30, 17, 4
    coefs = np.stack([synth_coefs(50 + i, bh, bw, q)[0] for i in range(n)])
    want = orc.dequant_idct_recon(coefs, q, bw, bh, n_planes=n).reshape(n, bh * 8, bw * 8)
    d_coefs = torch.from_numpy(coefs).cuda()
    d_pix = torch.zeros((n, bh * 8, bw * 8), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.dequant_idct_recon(d_coefs, q, bw, bh, n, d_pix)
    ctx.synchronize()
    ctx.reset_stream()
    assert np.array_equal(d_pix.cpu().numpy(), want)


def test_errors(ctx):
    import video_coding_amd as hvc
    q = np.ones(64, dtype=np.uint16)
    c = np.zeros((1, 1, 64), dtype=np.int16)
    out = np.zeros((8, 8), dtype=np.uint8)
    # a quantiser entry of zero: the decoder multiplies by it (decoder.ml:146 -- the model decodes such files), only the
    # encoder, which divides, refuses it (test_quantiser_entries_of_zero)
    with pytest.raises(hvc.HvcError) as e:
        ctx.dequant_idct_recon(c, q, 1, 1, 1, out, stride=12)
    assert e.value.code == -4
    with pytest.raises(hvc.HvcError) as e:
        ctx.dequant_idct_recon(c, q, 0, 1, 1, out)
    assert e.value.code == -1


def test_empty_batches_are_no_ops(ctx):
    """Zero planes / frames / files: every batch entry point returns HVC_OK and touches nothing."""
    import video_coding_amd as hvc
    q = np.ones(64, dtype=np.uint16)
    c = np.zeros((1, 1, 64), dtype=np.int16)
    out = np.full((8, 8), 0x77, dtype=np.uint8)
    ctx.dequant_idct_recon(c, q, 1, 1, 0, out)
    planes = [(2, 2, 0), (1, 1, 1), (1, 1, 1)]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    qt = np.ones((2, 64), dtype=np.uint16)
    coefs = np.zeros(cfs, dtype=np.int16)
    pix = np.full(pfs, 0x77, dtype=np.uint8)
    ctx.decode_frames(coefs, cfs, qt, specs, 0, pix, pfs)
    ctx.encode_frames(pix, pfs, qt, specs, 0, coefs, cfs)
    f444 = np.full(3 * 16 * 16, 0x77, dtype=np.uint8)
    ctx.decode_frames_yuv444(coefs, cfs, qt, specs, 0, 16, 16, f444)
    up = np.full((4, 4), 0x77, dtype=np.uint8)
    ctx.upsample420(np.zeros((2, 2), dtype=np.uint8), 2, 2, up, n_planes=0)
    st = ctx.jpeg_decode_batch([], pix, pfs)
    assert st.chunks == 0
    jpegs, st = ctx.jpeg_encode_batch([], 16, 16)
    assert jpegs == [] and st.chunks == 0
    assert (out == 0x77).all() and (pix == 0x77).all() and (f444 == 0x77).all() and (up == 0x77).all()
    assert not coefs.any()


def test_frame_count_limits(ctx):
    """grid.y carries the frame index, 65 535 per launch: a batch of any size is the library's to cut into launches, never
    the caller's to split -- 65 535, 65 536 and 70 000 one-block frames, host memory both ways."""
    import video_coding_amd as hvc
    n = 65535
    rng = np.random.Generator(np.random.PCG64(99))
    q = orc.quant_scale(orc.quant_luma(), 90).astype(np.uint16)
    coefs = rng.integers(-40, 41, size=(n, 64)).astype(np.int16)
    specs, cfs, pfs = hvc.hvc.frame_layout([(1, 1, 0)])
    pix = np.zeros((n, 64), dtype=np.uint8)
    ctx.decode_frames(coefs, cfs, q, specs, n, pix, pfs)
    want = orc.dequant_idct_recon(coefs.reshape(1, n, 64), q, n, 1).reshape(8, n, 8).transpose(1, 0, 2).reshape(n, 64)
    assert np.array_equal(pix, want)
    c1 = np.concatenate([coefs, coefs[:1][:, ::-1]])   # one frame more than a launch holds
    pix1 = np.zeros((n + 1, 64), dtype=np.uint8)
    ctx.decode_frames(c1, cfs, q, specs, n + 1, pix1, pfs)
    assert np.array_equal(pix1[:n], want)
    assert np.array_equal(pix1[n], orc.dequant_idct_recon(c1[n].reshape(1, 1, 64), q, 1, 1).reshape(64))
    # the per-plane entry point is the same call
    m = 70000
    c2 = rng.integers(-40, 41, size=(m, 64)).astype(np.int16)
    out = np.zeros((m, 64), dtype=np.uint8)
    ctx.dequant_idct_recon(c2, q, 1, 1, m, out)
    want2 = orc.dequant_idct_recon(c2.reshape(1, m, 64), q, m, 1).reshape(8, m, 8).transpose(1, 0, 2).reshape(m, 64)
    assert np.array_equal(out, want2)


def test_failed_call_between_wide_decodes_leaves_no_stale_fixups(ctx):
    """A decode with fix-up blocks, then a call that fails after its argument checks (misaligned device
    pointer), then a decode of a smaller geometry: the failed call must not hand the next one a
    fix-up counter that still holds the first call's count (its int64 kernel would re-process old
    block ids under the new geometry).  Device memory on both sides, so a stray write would land in the
    canary rows around the small plane."""
    import torch
    import video_coding_amd as hvc
    rng = np.random.Generator(np.random.PCG64(2025))
    q = np.full(64, 255, dtype=np.uint16)
    bw, bh = 64, 40                      # 2560 blocks, every one of them through the int64 kernel
    adv = rng.integers(-2047, 2048, size=(bh, bw, 64)).astype(np.int16)
    d_adv = torch.from_numpy(adv).cuda()
    d_big = torch.zeros((bh * 8, bw * 8), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        for fused_fail in (False, True):
            ctx.dequant_idct_recon(d_adv, q, bw, bh, 1, d_big)
            assert ctx.last_wide_blocks() > 0
            with pytest.raises(hvc.HvcError) as e:  # fails at the alignment check, after the geometry checks
                if fused_fail:
                    specs, cfs, _ = hvc.hvc.frame_layout([(2, 2, 0), (1, 1, 0), (1, 1, 0)])
                    ctx.decode_frames_yuv444(d_adv.data_ptr() + 2, cfs, q, specs, 1, 16, 16, d_big.data_ptr())
                else:
                    ctx.dequant_idct_recon(d_adv.data_ptr() + 2, q, bw, bh, 1, d_big.data_ptr())
            assert e.value.code == -4
            # smaller geometry, valid data, canary rows on both sides of the plane
            sbw, sbh = 3, 2
            small, _ = synth_coefs(77, sbh, sbw, q)
            d_small = torch.from_numpy(small).cuda()
            guard = 64
            d_pix = torch.full(((sbh * 8 + 2 * guard), sbw * 8), 0x5A, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctx.dequant_idct_recon(d_small, q, sbw, sbh, 1, d_pix[guard:guard + sbh * 8])
            ctx.synchronize()
            got = d_pix.cpu().numpy()
            assert ctx.last_wide_blocks() == 0
            assert np.array_equal(got[guard:guard + sbh * 8], orc.dequant_idct_recon(small, q, sbw, sbh).reshape(sbh * 8, sbw * 8))
            assert (got[:guard] == 0x5A).all() and (got[guard + sbh * 8:] == 0x5A).all()
    finally:
        ctx.reset_stream()


def test_a_million_blocks_within_one_per_cent_of_each_guard_limit(ctx):
    """The int32 kernel's guards (hvc_idct_spec.h: coefficient energy, row-output energy, the 181 * y arguments of
    the row and of the column pass) decide per block between the packed int32 path and the int64 kernel.  2 500
    random base blocks, each scaled so that ONE of the four guarded quantities lands at 100 different places within
    +-1 % of its limit: 10^6 blocks that sit right on the boundaries, on either side.  Every one must equal the
    model restatement (int64) -- whichever path it took -- and the int64-only kernel; both paths must have been
    taken."""
    from test_guard_bounds import idct_spec, model_pass
    c, _ = idct_spec()
    rng = np.random.Generator(np.random.PCG64(424242))
    zi = orc.zigzag_inverse()
    tables = [np.ones(64, np.uint16), orc.quant_scale(orc.quant_luma(), 75).astype(np.uint16),
              orc.quant_scale(orc.quant_chroma(), 30).astype(np.uint16), np.full(64, 255, np.uint16)]

    def guarded_quantities(block_zz, q):
        """(E, row-output energy, max |y| of the row passes, max |y| of the column passes) of one block, exact ints"""
        d = [0] * 64
        for k in range(64):
            d[zi[k]] = int(block_zz[k]) * int(q[k])

        def ys(b, col):  # the two arguments of the 181-products (dct.ml:41-42 / 83-84), via the butterfly's odd half
            r, s = (4, 3) if col else (0, 0)
            x4, x5, x6, x7 = b[1], b[7], b[5], b[3]
            t = 565 * (x4 + x5) + r
            x4, x5 = (t + (2841 - 565) * x4) >> s, (t - (2841 + 565) * x5) >> s
            t = 2408 * (x6 + x7) + r
            x6, x7 = (t - (2408 - 1609) * x6) >> s, (t - (2408 + 1609) * x7) >> s
            a, b2 = x4 - x6, x5 - x7
            return max(abs(a + b2), abs(a - b2))
        rows = [model_pass(d[8 * r:8 * r + 8], False) for r in range(8)]
        yr = max(ys(d[8 * r:8 * r + 8], False) for r in range(8))
        cols = [[rows[r][cc] for r in range(8)] for cc in range(8)]
        yc = max(ys(col, True) for col in cols)
        return (sum(int(v) ** 2 for v in block_zz), sum(v * v for row in rows for v in row), yr, yc)

    n_base, n_eps = 2500, 100
    eps = np.linspace(-0.01, 0.01, n_eps)
    blocks, qsel = [], []
    for i in range(n_base):
        ti = i % len(tables)
        q = tables[ti]
        dens = (1, 4, 16, 64)[(i // 4) % 4]   # from a lone coefficient to a dense block
        base = np.zeros(64, np.int64)
        idx = rng.choice(64, size=dens, replace=False)
        base[idx] = rng.integers(-200, 201, size=dens)
        base[idx] += np.where(base[idx] == 0, 7, 0)
        E1, R1, Yr1, Yc1 = guarded_quantities(base, q)
        qmax = int(q.max())
        limits = ((E1, (c["HVC_GUARD_D_PACKED"] // qmax) ** 2, 0.5), (R1, c["HVC_GUARD_RE"], 0.5),
                  (Yr1, c["HVC_GUARD_Y"], 1.0), (Yc1, c["HVC_GUARD_Y"], 1.0))
        for val, lim, power in limits:
            s0 = (lim / max(val, 1)) ** power
            for e in eps:   # quantity ~ s^(1/power): (1 + e) on the quantity is (1 + e)^power on the scale
                blocks.append(np.clip(np.rint(base * s0 * (1.0 + e) ** power), -32768, 32767).astype(np.int16))
                qsel.append(ti)
    blocks = np.stack(blocks)
    qsel = np.array(qsel)
    assert blocks.shape[0] == 1_000_000
    took_wide = took_fast = 0
    for ti, q in enumerate(tables):
        sel = blocks[qsel == ti]
        n = sel.shape[0]
        assert n == 250_000
        coefs = np.ascontiguousarray(sel.reshape(500, 500, 64))
        want = orc.dequant_idct_recon(coefs, q, 500, 500).reshape(4000, 4000)
        got = gpu_decode_plane(ctx, coefs, q)
        wide = ctx.last_wide_blocks()
        assert np.array_equal(got, want), ti
        took_wide += wide
        took_fast += n - wide
        ctx.set_decode_kernel(2)
        try:
            assert np.array_equal(gpu_decode_plane(ctx, coefs, q), want), ti
        finally:
            ctx.set_decode_kernel(0)
    # the boundaries really were straddled.  (Scaled up to the limit of ONE guard a block has usually tripped another
    # one already -- the coefficient energy binds first for most tables -- so the int64 side is the larger one: about
    # 9 in 10; the blocks aimed at a block's binding guard land on both sides of it.)
    assert took_wide > 50_000 and took_fast > 50_000, (took_wide, took_fast)


def test_batches_cut_into_several_launches_give_the_same_bytes():
    """Device-memory batches above HVC_LAUNCH_BYTES (default 10 GB of algorithmic bytes) are cut into equal launches
    (hvc_capi.hip launch_bytes_limit: launches beyond ~3 ms run 2-3 % slower).  With the limit set to 150 kB a 23-frame
    batch becomes a dozen launches: decode (with blocks for the fix-up list in several parts), fused 4:4:4 and encode
    must give the bytes of the one-launch form.  The variable is read once per process, hence the child process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np, torch
from helpers import synth_pixels
from oracle import orc
import video_coding_amd as hvc
c = hvc.Context(0)
c.set_stream(torch.cuda.current_stream().cuda_stream)
planes = [(20, 12, 0), (10, 6, 1), (10, 6, 1)]
specs, cfs, pfs = hvc.hvc.frame_layout(planes)
qt = np.stack([orc.quant_scale(orc.quant_luma(), 75), orc.quant_scale(orc.quant_chroma(), 75)]).astype(np.uint16)
n = 23
rng = np.random.Generator(np.random.PCG64(3))
pix = np.stack([np.concatenate([synth_pixels(100 * f + i, bh * 8, bw * 8).reshape(-1) for i, (bw, bh, _) in enumerate(planes)]) for f in range(n)])
coefs = np.zeros((n, cfs), np.int16)
want_pix = np.zeros((n, pfs), np.uint8)
for f in range(n):
    for s, (bw, bh, q) in zip(specs, planes):
        p = pix[f, s["plane_offset"]:s["plane_offset"] + bw * bh * 64].reshape(bh * 8, bw * 8)
        cf = orc.fdct_quant(p, qt[q], bw, bh).reshape(-1)
        if f % 5 == 2:   # adversarial blocks: the fix-up list has entries in several of the launches
            cf = rng.integers(-2047, 2048, size=cf.size).astype(np.int16)
        coefs[f, s["coef_offset"]:s["coef_offset"] + cf.size] = cf
        want_pix[f, s["plane_offset"]:s["plane_offset"] + bw * bh * 64] = orc.dequant_idct_recon(cf, qt[q], bw, bh)
d_c = torch.from_numpy(coefs).cuda()
d_p = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
c.decode_frames(d_c, cfs, qt, specs, n, d_p, pfs)
c.synchronize()
assert np.array_equal(d_p.cpu().numpy(), want_pix), "decode"
# hvc_last_wide_blocks covers the whole call, not its last launch (ADVICE r2): the total over the split launches equals
# the count of the frame-by-frame form, and the fused entry point agrees with itself the same way
wide_call = c.last_wide_blocks()
wide_sum = 0
for f in range(n):
    c.decode_frames(d_c[f:f + 1], cfs, qt, specs, 1, d_p[f:f + 1], pfs)
    wide_sum += c.last_wide_blocks()
assert wide_call == wide_sum > 0, (wide_call, wide_sum)
assert wide_call > c.last_wide_blocks()  # (the last frame alone has fewer: a per-launch count would have said that)
print("wide blocks of the split call", wide_call)
# encode
d_x = torch.from_numpy(pix).cuda()
d_o = torch.zeros((n, cfs), dtype=torch.int16, device="cuda")
c.encode_frames(d_x, pfs, qt, specs, n, d_o, cfs)
c.synchronize()
got = d_o.cpu().numpy()
for f in range(n):
    for s, (bw, bh, q) in zip(specs, planes):
        p = pix[f, s["plane_offset"]:s["plane_offset"] + bw * bh * 64].reshape(bh * 8, bw * 8)
        assert np.array_equal(got[f, s["coef_offset"]:s["coef_offset"] + bw * bh * 64], orc.fdct_quant(p, qt[q], bw, bh).reshape(-1)), ("encode", f)
# fused 4:4:4 against the host-memory form (one part: below 64 MB it is a single launch... of the same entry point)
W, H = 160, 96
d_f = torch.zeros((n, 3 * W * H), dtype=torch.uint8, device="cuda")
c.decode_frames_yuv444(d_c, cfs, qt, specs, n, W, H, d_f)
c.synchronize()
for f in range(n):
    y = want_pix[f, :160 * 96].reshape(96, 160)
    u = want_pix[f, 160 * 96:160 * 96 + 80 * 48].reshape(48, 80)
    v = want_pix[f, 160 * 96 + 80 * 48:].reshape(48, 80)
    want = np.concatenate([y.reshape(-1), orc.supersample_hv2(u).reshape(-1), orc.supersample_hv2(v).reshape(-1)])
    assert np.array_equal(d_f[f].cpu().numpy(), want), ("444", f)
wide_call = c.last_wide_blocks()
wide_sum = 0
for f in range(n):
    c.decode_frames_yuv444(d_c[f:f + 1], cfs, qt, specs, 1, W, H, d_f[f:f + 1])
    wide_sum += c.last_wide_blocks()
assert wide_call == wide_sum > 0, ("444", wide_call, wide_sum)
print("split ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVC_LAUNCH_BYTES="150000")  # 69 KB a frame: two frames per launch
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "split ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_quantiser_entries_of_zero(ctx):
    """A DQT may hold zeros (no encoder writes them; a mutated file does): `dequant.(i) <- coefs.(i) * qnt_tab.(i)`
    (decoder.ml:146) makes the coefficient 0 and the model decodes the file -- so does the block stage, in its packed,
    plain and int64 forms; the encoder's quantiser divides (encoder.ml:98-101: Division_by_zero) and refuses the table."""
    import video_coding_amd as hvc
    rng = np.random.Generator(np.random.PCG64(77))
    bw, bh = 9, 5
    for amp in (40, 2000):          # small coefficients (packed kernel) and large ones (fix-up path)
        coefs = rng.integers(-amp, amp + 1, size=(bh, bw, 64)).astype(np.int16)
        q = rng.integers(1, 60, size=64).astype(np.uint16)
        q[[0, 3, 17, 63]] = 0
        out = np.zeros((bh * 8, bw * 8), dtype=np.uint8)
        ctx.dequant_idct_recon(coefs, q, bw, bh, 1, out)
        want = orc.dequant_idct_recon(coefs, q, bw, bh)
        assert np.array_equal(out, np.asarray(want).reshape(out.shape)), amp
    # whole file: mini.jpg with zeros written into its chroma DQT
    data = bytearray(golden_bytes("mini.jpg"))
    i = data.index(b"\xff\xdb", data.index(b"\xff\xdb") + 2)
    for k in (5, 9, 40, 64):
        data[i + 4 + k] = 0
    data = bytes(data)
    info, pixels = ctx.jpeg_decode(data)
    d = orc.Decoder(data)
    d.decode()
    for k, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, d.plane(k)), k
    # the encoder refuses
    px = rng.integers(0, 256, size=(16, 16), dtype=np.uint8)
    q0 = np.full(64, 7, dtype=np.uint16)
    q0[9] = 0
    with pytest.raises(hvc.HvcError) as e:
        ctx.fdct_quant(px, q0, 2, 2, 1, np.zeros((2, 2, 64), dtype=np.int16))
    assert e.value.code == -5
