"""hvc_decode_frames_yuv444: the block stage, Decoder.get_yuv_frame's crop (decoder.ml:403-420) and
Planar_444.convert_from_420 (tools/src/planar_444.ml:82-131) fused in one pass on the GPU, against
the CPU oracle composed the way the reference composes them:
    decode planes -> crop to actual size -> supersample_hv2 on the cropped chroma planes."""
import numpy as np
import pytest

from helpers import synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def geometry420(width, height):
    """Decoder.init for a 4:2:0 scan (decoder.ml:304-345): planes rounded to the 16 x 16 MCU."""
    rw, rh = (width + 15) // 16 * 16, (height + 15) // 16 * 16
    return [(rw // 8, rh // 8, 0), (rw // 16, rh // 16, 1), (rw // 16, rh // 16, 1)]


def make_record(seed, planes, qtabs, adversarial=0.0):
    """One frame's coefficient record (valid blocks from the oracle's forward path; a fraction of
    blocks replaced by dense +-2047 coefficients that leave the int32 kernel's proven range)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rec = []
    for i, (bw, bh, qt) in enumerate(planes):
        c = orc.fdct_quant(synth_pixels(seed * 8 + i, bh * 8, bw * 8), qtabs[qt], bw, bh).reshape(bh * bw, 64)
        if adversarial:
            pick = rng.random(bh * bw) < adversarial
            c[pick] = rng.choice(np.array([-2047, 2047, -1024, 1023], dtype=np.int16), size=(int(pick.sum()), 64))
        rec.append(c.reshape(-1))
    return np.concatenate(rec)


def expected444(rec, planes, qtabs, width, height):
    out, off = [], 0
    for i, (bw, bh, qt) in enumerate(planes):
        n = bw * bh * 64
        plane = orc.dequant_idct_recon(rec[off:off + n], qtabs[qt], bw, bh).reshape(bh * 8, bw * 8)
        off += n
        if i == 0:
            out.append(plane[:height, :width])
        else:
            out.append(orc.supersample_hv2(np.ascontiguousarray(plane[:height // 2, :width // 2])))
    return np.concatenate([p.reshape(-1) for p in out])


def tables(quality=75):
    return np.stack([orc.quant_scale(orc.quant_luma(), quality), orc.quant_scale(orc.quant_chroma(), quality)]).astype(np.uint16)


def run(ctx, recs, planes, qtabs, width, height, device, frame_stride=None):
    import torch
    import video_coding_amd as hvc
    specs, cfs, _ = hvc.hvc.frame_layout(planes)
    n = len(recs)
    fs = frame_stride or 3 * width * height
    coefs = np.stack(recs)
    if device:
        d_c = torch.from_numpy(coefs).cuda()
        d_o = torch.full((n * fs,), 0xA5, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_frames_yuv444(d_c, cfs, qtabs, specs, n, width, height, d_o, fs)
        ctx.synchronize()
        return d_o.cpu().numpy().reshape(n, fs)
    out = np.full((n, fs), 0xA5, dtype=np.uint8)
    ctx.decode_frames_yuv444(coefs, cfs, qtabs, specs, n, width, height, out, fs)
    return out


@pytest.mark.parametrize("width,height", [
    (64, 48),      # one tile, block aligned
    (64, 40),      # chroma crop ends inside a block row (ah = 20)
    (1056, 144),   # chroma 528 x 72: two tiles across (sharing block column 63), three down (seams at rows 31, 63)
    (2080, 80),    # chroma 130 blocks wide: three tiles across
    (1024, 32),    # chroma exactly 64 blocks wide: one tile, lane 63 owns the last column
    (1040, 32),    # 65 blocks: the second tile holds two columns
    (2032, 32),    # 127 blocks = 2 * 63 + 1: the last column is the second tile's overlap lane
    (48, 272),     # chroma 136 rows: five tiles down
    (1920, 1080),  # the headline geometry: 1088 decoded rows, crop at 1080 / 540
])
@pytest.mark.parametrize("device", [True, False])
def test_fused_444_equals_decode_crop_upsample(ctx, width, height, device):
    planes, qt = geometry420(width, height), tables()
    recs = [make_record(3 + f, planes, qt) for f in range(2)]
    got = run(ctx, recs, planes, qt, width, height, device)
    for f, rec in enumerate(recs):
        want = expected444(rec, planes, qt, width, height)
        assert np.array_equal(got[f], want), (f, int(np.flatnonzero(got[f] != want)[0]))
    assert ctx.last_wide_blocks() == 0


@pytest.mark.parametrize("width,height", [(52, 44), (100, 30), (18, 10), (2, 2), (1042, 70), (24, 8)])
def test_unaligned_sizes_take_the_byte_path(ctx, width, height):
    """width % 16 != 0: rows are not 16-byte aligned, the kernel stores bytes with bounds checks."""
    planes, qt = geometry420(width, height), tables(60)
    recs = [make_record(11 + f, planes, qt) for f in range(3)]
    got = run(ctx, recs, planes, qt, width, height, True)
    for f, rec in enumerate(recs):
        assert np.array_equal(got[f], expected444(rec, planes, qt, width, height)), f


def test_frame_stride_padding_is_left_untouched(ctx):
    width, height = 64, 32
    planes, qt = geometry420(width, height), tables()
    recs = [make_record(21 + f, planes, qt) for f in range(3)]
    fs = 3 * width * height + 48
    got = run(ctx, recs, planes, qt, width, height, True, frame_stride=fs)
    for f, rec in enumerate(recs):
        assert np.array_equal(got[f][:3 * width * height], expected444(rec, planes, qt, width, height))
        assert (got[f][3 * width * height:] == 0xA5).all()


@pytest.mark.parametrize("width,height", [(1056, 144), (80, 48), (52, 44)])
def test_guard_failures_go_through_the_wide_kernel_and_the_reinterpolation(ctx, width, height):
    """Blocks outside the int32 kernel's proven range: int64 kernel writes the source samples, the
    third pass rebuilds every interpolated sample that reads them (incl. the neighbours' edges)."""
    planes, qt = geometry420(width, height), tables(90)
    recs = [make_record(31 + f, planes, qt, adversarial=0.07) for f in range(2)]
    got = run(ctx, recs, planes, qt, width, height, True)
    assert ctx.last_wide_blocks() > 0
    for f, rec in enumerate(recs):
        want = expected444(rec, planes, qt, width, height)
        assert np.array_equal(got[f], want), (f, int(np.flatnonzero(got[f] != want)[0]))


def test_all_blocks_adversarial(ctx):
    width, height = 1056, 80
    planes, qt = geometry420(width, height), tables(100)
    recs = [make_record(41, planes, qt, adversarial=1.0)]
    got = run(ctx, recs, planes, qt, width, height, True)
    assert np.array_equal(got[0], expected444(recs[0], planes, qt, width, height))


def test_three_implementations_agree(ctx):
    """packed + fix-up, wide-only (int64 for every block) and 16-bit tables (wide-only by rule)."""
    width, height = 1056, 144
    planes, qt = geometry420(width, height), tables(50)
    recs = [make_record(51 + f, planes, qt, adversarial=0.02) for f in range(2)]
    a = run(ctx, recs, planes, qt, width, height, True)
    ctx.set_decode_kernel(2)
    try:
        b = run(ctx, recs, planes, qt, width, height, True)
    finally:
        ctx.set_decode_kernel(0)
    assert np.array_equal(a, b)
    for f, rec in enumerate(recs):
        assert np.array_equal(a[f], expected444(rec, planes, qt, width, height))
    qt16 = qt.copy()
    qt16[0, 63] = 300  # an entry above 255 sends the whole call to the int64 kernel
    c = run(ctx, recs[:1], planes, qt16, width, height, True)
    assert np.array_equal(c[0], expected444(recs[0], planes, qt16, width, height))


def test_equals_the_separate_kernels_at_full_size(ctx):
    """1080p x 12 frames: fused output == hvc_decode_frames -> crop -> hvc_upsample420 (product path,
    three kernels) on every frame; frame 0 also against the oracle."""
    import torch
    import video_coding_amd as hvc
    width, height = 1920, 1080
    planes, qt = geometry420(width, height), tables()
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    base = [make_record(61 + f, planes, qt) for f in range(3)]
    n = 12
    d_c = torch.from_numpy(np.stack(base)).cuda().repeat(4, 1).contiguous()
    d_o = torch.zeros((n, 3 * width * height), dtype=torch.uint8, device="cuda")
    d_p = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.decode_frames_yuv444(d_c, cfs, qt, specs, n, width, height, d_o)
    ctx.decode_frames(d_c, cfs, qt, specs, n, d_p, pfs)
    ctx.synchronize()
    ref = torch.zeros_like(d_o)
    ref[:, :width * height] = d_p[:, :planes[0][0] * 8 * planes[0][1] * 8].reshape(n, -1, planes[0][0] * 8)[:, :height, :width].reshape(n, -1)
    cw, ch = width // 2, height // 2
    for i in (1, 2):
        off = specs[i]["plane_offset"]
        pw, ph = planes[i][0] * 8, planes[i][1] * 8
        src = d_p[:, off:off + pw * ph].reshape(n, ph, pw)[:, :ch, :cw].contiguous()
        dst = torch.zeros((n, height, width), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.upsample420(src, cw, ch, dst, n_planes=n)
        ctx.synchronize()
        ref[:, i * width * height:(i + 1) * width * height] = dst.reshape(n, -1)
    assert torch.equal(d_o, ref)
    assert np.array_equal(d_o[0].cpu().numpy(), expected444(base[0], planes, qt, width, height))


def test_argument_errors(ctx):
    import video_coding_amd as hvc
    planes, qt = geometry420(64, 48), tables()
    specs, cfs, _ = hvc.hvc.frame_layout(planes)
    coefs = np.zeros(cfs, dtype=np.int16)
    out = np.zeros(3 * 64 * 48, dtype=np.uint8)
    with pytest.raises(hvc.HvcError):  # odd width: "Expecting a 4:2:0 frame" (yuv.ml:104-116)
        ctx.decode_frames_yuv444(coefs, cfs, qt, specs, 1, 63, 48, out)
    with pytest.raises(hvc.HvcError):  # crop outside the decoded planes
        ctx.decode_frames_yuv444(coefs, cfs, qt, specs, 1, 64, 66, out)
    with pytest.raises(hvc.HvcError):  # not three components
        ctx.decode_frames_yuv444(coefs, cfs, qt, specs[:2], 1, 64, 48, out)


def test_single_frame_ignores_the_frame_stride(ctx):
    import video_coding_amd as hvc
    width, height = 48, 32
    planes, qt = geometry420(width, height), tables()
    rec = make_record(71, planes, qt)
    specs, cfs, _ = hvc.hvc.frame_layout(planes)
    out = np.zeros(3 * width * height, dtype=np.uint8)
    ctx.decode_frames_yuv444(rec, cfs, qt, specs, 1, width, height, out, 16)  # host buffers, stride < frame
    assert np.array_equal(out, expected444(rec, planes, qt, width, height))


@pytest.mark.parametrize("pad", [0, 1000])
def test_host_buffers_in_four_overlapped_parts_equal_the_device_path(ctx, pad):
    """hvc_decode_frames_yuv444 with host memory, a batch above 64 MB of coefficients (uploaded, decoded and
    downloaded in four parts on two streams): same frames as the resident path, the caller's bytes between
    frames untouched."""
    import torch
    import video_coding_amd as hvc
    width, height = 1920, 1080
    planes, qt = geometry420(width, height), tables()
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    base = [make_record(161 + f, planes, qt) for f in range(3)]
    n = 24
    batch = np.ascontiguousarray(np.stack(base)[np.arange(n) % 3])
    d_o = torch.zeros((n, 3 * width * height), dtype=torch.uint8, device="cuda")
    ctx.decode_frames_yuv444(torch.from_numpy(batch).cuda(), cfs, qt, specs, n, width, height, d_o)
    ctx.synchronize()
    want = d_o.cpu().numpy()
    fs = 3 * width * height + pad
    host = np.full((n, fs), 0x3C, dtype=np.uint8)
    ctx.decode_frames_yuv444(batch, cfs, qt, specs, n, width, height, host, frame_stride=fs)
    assert np.array_equal(host[:, :3 * width * height], want)
    assert (host[:, 3 * width * height:] == 0x3C).all()


def test_split_launch_modes_give_the_same_frames():
    """HVC_444_MODE=1 / 2 (hvc_capi.hip fused444_mode: the luma planes through k_decode_packed itself, before or beside the
    chroma tiles' kernel) are A/B alternates of the one-kernel form that ships (measured slower: profiles/r03b_ab.txt);
    the variable is read once per process, so this module runs again in child processes under both."""
    import os
    import subprocess
    import sys
    if os.environ.get("HVC_444_MODE"):
        pytest.skip("already inside a child run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("1", "2"):
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu"], cwd=root,
                           env=dict(os.environ, HVC_444_MODE=mode), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and " passed" in r.stdout, (mode, r.stdout[-2000:], r.stderr[-2000:])
