"""The asynchronous seam of include/hvc_jpeg.h -- hvc_host_alloc / hvc_host_register, hvc_decode_frames_submit,
hvc_encode_frames_submit, hvc_wait ("submit(frame batch, stream slot) / wait(slot)") -- against the oracle and
the blocking entry points: the same arithmetic (decoder.ml:142-149, 213-224; dct.ml:11-107; encoder.ml:81-108), only when the
caller gets its thread back differs.  And BASELINE config 3's shape driven from the caller's side: 4096 x 1080p coefficient
records through pinned slots, every decoded record K5-verified."""
import os
import sys
import threading

import numpy as np
import pytest

from helpers import synth_coefs, synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture()
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def _frame_set(n, planes, seed):
    """n frames of `planes` [(bw, bh, qtab)]: coefficient records (oracle forward path), the oracle's pixel records, tables"""
    import video_coding_amd as hvc
    q = np.stack([orc.quant_scale(orc.quant_luma(), 75), orc.quant_scale(orc.quant_chroma(), 75)]).astype(np.uint16)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    coefs = np.zeros((n, cfs), dtype=np.int16)
    want = np.zeros((n, pfs), dtype=np.uint8)
    pix = np.zeros((n, pfs), dtype=np.uint8)
    for f in range(n):
        for s in specs:
            bw, bh = s["blocks_w"], s["blocks_h"]
            c, p = synth_coefs(seed + 17 * f + s["coef_offset"] % 97, bh, bw, q[s["qtab"]])
            coefs[f, s["coef_offset"]:s["coef_offset"] + c.size] = c.reshape(-1)
            pix[f, s["plane_offset"]:s["plane_offset"] + p.size] = p.reshape(-1)
            want[f, s["plane_offset"]:s["plane_offset"] + p.size] = orc.dequant_idct_recon(c, q[s["qtab"]], bw, bh).reshape(-1)
    return specs, cfs, pfs, q, coefs, want, pix


def test_submit_wait_matches_the_oracle_host_and_device_output(ctx):
    import torch
    import video_coding_amd as hvc
    planes = [(12, 10, 0), (6, 5, 1), (6, 5, 1)]
    n = 5
    specs, cfs, pfs, q, coefs, want, _ = _frame_set(n, planes, 900)
    comps = hvc.hvc.components(specs)
    pin_c = ctx.host_alloc((n, cfs), np.int16)
    pin_p = ctx.host_alloc((n, pfs), np.uint8)
    try:
        pin_c[:] = coefs
        pin_p[:] = 0xEE
        ctx.decode_frames_submit(0, pin_c, cfs, q, comps, n, pin_p, pfs)           # pinned in, pinned out
        d_p = torch.full((n, pfs), 0x11, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.decode_frames_submit(1, pin_c, cfs, q, comps, n, d_p, pfs)             # pinned in, device out
        page_p = np.zeros((n, pfs), dtype=np.uint8)
        ctx.decode_frames_submit(2, coefs, cfs, q, comps, n, page_p, pfs)          # pageable both ways: same result
        ctx.wait(2)
        ctx.wait(0)
        ctx.wait(1)
        assert np.array_equal(pin_p, want) and np.array_equal(page_p, want) and np.array_equal(d_p.cpu().numpy(), want)
        st = ctx.slot_last_stats(0)
        assert st.h2d_bytes == coefs.nbytes and st.d2h_bytes == want.size and st.h2d_ms > 0 and st.kernel_ms > 0 and st.d2h_ms > 0
        assert ctx.slot_last_stats(1).d2h_bytes == 0
        assert ctx.slot_done(0) and ctx.slot_done(3)
    finally:
        ctx.host_free(pin_c)
        ctx.host_free(pin_p)


def test_slot_protocol_and_errors(ctx):
    import video_coding_amd as hvc
    planes = [(4, 3, 0)]
    specs, cfs, pfs, q, coefs, want, _ = _frame_set(2, planes, 77)
    comps = hvc.hvc.components(specs)
    out = np.zeros((2, pfs), dtype=np.uint8)
    L = hvc.lib()
    ctx.wait(0)                                                 # an idle slot: nothing to wait for
    with pytest.raises(hvc.HvcError) as e:
        ctx.slot_last_stats(0)                                  # ... and nothing completed in it yet
    assert e.value.code == -1
    ctx.decode_frames_submit(0, coefs, cfs, q, comps, 2, out, pfs)
    with pytest.raises(hvc.HvcError) as e:                      # the slot is taken until its wait
        ctx.decode_frames_submit(0, coefs, cfs, q, comps, 2, out, pfs)
    assert e.value.code == hvc.hvc.HVC_E_BUSY and b"hvc_wait" in L.hvc_strerror(hvc.hvc.HVC_E_BUSY)
    ctx.wait(0)
    assert np.array_equal(out, want)
    for slot in (-1, hvc.hvc.HVC_SLOTS):
        with pytest.raises(hvc.HvcError) as e:
            ctx.decode_frames_submit(slot, coefs, cfs, q, comps, 2, out, pfs)
        assert e.value.code == -1
        assert L.hvc_wait(ctx._h, slot) == -1
    # what hvc_decode_frames refuses, the submission refuses at once, and the slot stays free
    with pytest.raises(hvc.HvcError) as e:
        ctx.decode_frames_submit(1, coefs, cfs - 8, q, comps, 2, out, pfs)       # frame stride shorter than the record
    assert e.value.code == -1
    with pytest.raises(hvc.HvcError) as e:
        ctx.decode_frames_submit(1, coefs, cfs, q, comps, 2, out, pfs + 4)       # stride not a multiple of 8
    assert e.value.code == -4
    ctx.decode_frames_submit(1, coefs, cfs, q, comps, 0, out, pfs)               # no frames: nothing in flight
    assert ctx.slot_done(1)
    ctx.decode_frames_submit(1, coefs, cfs, q, comps, 2, out, pfs)
    ctx.wait(1)


def test_submit_padded_strides_and_untouched_padding(ctx):
    """planes with a row stride beyond their width and gaps between frames: only the planes' bytes come back"""
    import video_coding_amd as hvc
    bw, bh, n = 7, 5, 3
    q = orc.quant_scale(orc.quant_luma(), 50).astype(np.uint16)
    stride, gap = bw * 8 + 24, 4096
    pfs = stride * bh * 8 + gap
    cfs = bw * bh * 64 + 64
    specs = [dict(blocks_w=bw, blocks_h=bh, qtab=0, coef_offset=0, plane_offset=0, stride=stride)]
    coefs = np.zeros((n, cfs), dtype=np.int16)
    out = np.full((n, pfs), 0xA5, dtype=np.uint8)
    want = out.copy()
    for f in range(n):
        c, _ = synth_coefs(40 + f, bh, bw, q)
        coefs[f, :c.size] = c.reshape(-1)
        rows = want[f, :stride * bh * 8].reshape(bh * 8, stride)
        rows[:, :bw * 8] = orc.dequant_idct_recon(c, q, bw, bh).reshape(bh * 8, bw * 8)
    ctx.decode_frames_submit(3, coefs, cfs, q, hvc.hvc.components(specs), n, out, pfs)
    ctx.wait(3)
    assert np.array_equal(out, want)


def test_encode_submit_matches_the_blocking_call_and_the_oracle(ctx):
    import torch
    import video_coding_amd as hvc
    planes = [(10, 6, 0), (5, 3, 1), (5, 3, 1)]
    n = 4
    specs, cfs, pfs, q, coefs, _, pix = _frame_set(n, planes, 333)
    comps = hvc.hvc.components(specs)
    host_c = np.zeros((n, cfs), dtype=np.int16)
    d_c = torch.zeros((n, cfs), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_frames_submit(0, pix, pfs, q, comps, n, host_c, cfs)
    ctx.encode_frames_submit(1, pix, pfs, q, comps, n, d_c, cfs)
    ctx.wait(0)
    ctx.wait(1)
    assert np.array_equal(host_c, coefs) and np.array_equal(d_c.cpu().numpy(), coefs)   # (coefs: the oracle's forward path)


def test_registered_caller_memory_and_a_filling_thread(ctx):
    """the caller's own page-aligned arrays pinned in place (hvc_host_register: whole pages only), all slots in flight, a second
    thread writing the NEXT batch's records while the GPU works: every batch equals the oracle's frames"""
    import video_coding_amd as hvc
    planes = [(16, 12, 0), (8, 6, 1), (8, 6, 1)]
    n, batches = 6, 9
    specs, cfs, pfs, q, coefs, want, _ = _frame_set(n, planes, 1234)
    comps = hvc.hvc.components(specs)
    S = hvc.hvc.HVC_SLOTS
    bufs = [hvc.hvc.page_aligned_empty((n, cfs), np.int16) for _ in range(S)]     # whole pages of their own: what may be registered
    outs = [hvc.hvc.page_aligned_empty((n, pfs), np.uint8) for _ in range(S)]
    for a in bufs + outs:
        a[:] = 0
        ctx.host_register(a)
    # memory that shares its pages with whatever malloc put beside it is refused (the runtime finds registered memory by page)
    odd = np.zeros(3 * hvc.hvc.PAGE, dtype=np.uint8)
    assert hvc.lib().hvc_host_register(ctx._h, odd.ctypes.data + 8, hvc.hvc.PAGE) == -4
    assert hvc.lib().hvc_host_register(ctx._h, (odd.ctypes.data + hvc.hvc.PAGE - 1) // hvc.hvc.PAGE * hvc.hvc.PAGE, 100) == -4
    try:
        # batch k holds the frames rotated by k: a stale buffer would show
        def fill(s, k):
            bufs[s][:] = np.roll(coefs, k, axis=0)

        fill(0, 0)
        for k in range(batches):
            s = k % S
            t = None
            if k + 1 < batches:
                s2 = (k + 1) % S
                if k + 1 >= S:                       # the next slot's previous occupant: retire it, check it
                    ctx.wait(s2)
                    assert np.array_equal(outs[s2], np.roll(want, k + 1 - S, axis=0)), k
                t = threading.Thread(target=fill, args=(s2, k + 1))
                t.start()                            # ... refilled while batch k is submitted and runs
            ctx.decode_frames_submit(s, bufs[s], cfs, q, comps, n, outs[s], pfs)
            if t:
                t.join()
        for k in range(max(0, batches - S), batches):
            ctx.wait(k % S)
            assert np.array_equal(outs[k % S], np.roll(want, k, axis=0)), k
    finally:
        for a in bufs + outs:
            ctx.host_unregister(a)


def test_destroy_with_a_submission_in_flight():
    """hvc_destroy drains what nobody waited for"""
    import video_coding_amd as hvc
    planes = [(32, 32, 0)]
    specs, cfs, pfs, q, coefs, want, _ = _frame_set(3, planes, 5)
    c = hvc.Context(0)
    out = np.zeros((3, pfs), dtype=np.uint8)
    c.decode_frames_submit(0, coefs, cfs, q, hvc.hvc.components(specs), 3, out, pfs)
    c.close()
    assert np.array_equal(out, want)


def test_4096_records_through_pinned_slots_every_record_verified():
    """4096 x 1080p coefficient records from pinned slots, caller threads refilling slot k + 1 during slot
    k, every decoded record K5-verified against tests/golden/bench_checksums.json (configs_c3); and the same with the pixel
    records coming back into pinned host memory, every one compared byte for byte"""
    import bench_configs as bc
    r = bc.config_async(bc.make_args(frames=4096, steps=1, threads=16, chunk=64))
    assert r["frames"] == 4096 and r["checksum"]["records"] == 4096 and r["checksum"]["verified"] is True, r
    assert r["h2d_GBps"] > 5 and 0.0 < r["overlap_fraction"] < 1.0 and r["value"] > 2000.0, r
    r = bc.config_async(bc.make_args(frames=1024, steps=1, threads=16, chunk=64, host_out=True))
    assert r["checksum"]["records"] == 1024 and r["checksum"]["verified"] is True, r
    assert r["d2h_GBps"] > 2 and r["value"] > 2000.0, r


def test_a_batch_beyond_the_grid_is_cut_not_refused(ctx):
    """70 000 frames of 8 x 8-block planes (more than a grid's 65535 rows) decode in one call, K5 takes
    70 000 records in one call, and the encoder side does the same"""
    import torch
    import video_coding_amd as hvc
    from helpers import checksum_records
    bw = bh = 8
    n, D = 70000, 7
    q = orc.quant_scale(orc.quant_luma(), 75).astype(np.uint16)
    dist = np.stack([synth_coefs(4000 + i, bh, bw, q)[0].reshape(-1) for i in range(D)])
    want = checksum_records(np.stack([orc.dequant_idct_recon(dist[i].reshape(bh, bw, 64), q, bw, bh).reshape(-1) for i in range(D)]))
    d_c = torch.from_numpy(dist).cuda().repeat((n + D - 1) // D, 1)[:n].contiguous()
    d_p = torch.zeros((n, bw * bh * 64), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.dequant_idct_recon(d_c, q, bw, bh, n, d_p)
        sums = ctx.checksum_records(d_p, bw * bh * 64, n)
        assert sums.shape == (n,) and np.array_equal(sums, want[np.arange(n) % D])
        assert ctx.last_wide_blocks() == 0
        # the other direction: the decoded planes back through the forward path, one call
        d_c2 = torch.zeros_like(d_c)
        ctx.fdct_quant(d_p, q, bw, bh, n, d_c2)
        ctx.synchronize()
        first = d_c2[:D].cpu().numpy()
        assert bool((d_c2.view(-1, D, bw * bh * 64)[:n // D] == d_c2[:D][None]).all())
        for i in range(D):
            pl = orc.dequant_idct_recon(dist[i].reshape(bh, bw, 64), q, bw, bh).reshape(bh * 8, bw * 8)
            assert np.array_equal(first[i], orc.fdct_quant(pl, q, bw, bh).reshape(-1))
        # every block through the int64 kernel, cut the same way
        ctx.set_decode_kernel(2)
        d_p.zero_()
        ctx.dequant_idct_recon(d_c, q, bw, bh, n, d_p)
        assert np.array_equal(ctx.checksum_records(d_p, bw * bh * 64, n), want[np.arange(n) % D])
        assert ctx.last_wide_blocks() == n * bw * bh
    finally:
        ctx.set_decode_kernel(0)
        ctx.reset_stream()


def test_wide_kernel_bench_modes_verify():
    """the int64 kernel as a whole call (hvc_set_decode_kernel 2, and a DQT entry above 255), timed and equal to
    the packed kernel's frames"""
    import bench_configs as bc
    for mode in ("kernel2", "dqt16"):
        r = bc.config_wide(bc.make_args(frames=64, steps=3, warmup=2, wide_mode=mode))
        assert r["checksum"]["verified"] is True and r["wide_path_blocks"] == r["all_blocks"], r
        assert r["frac_of_8TBps"] > 0.02, r       # (the scratch-memory kernel of round 5 sat far below this)


def test_randomised_submissions_interleaved_with_the_blocking_entry_points(ctx):
    """Slots in any order, decode and encode submissions mixed, host and device destinations, pinned and pageable sources, other
    entry points of the SAME context in between (blocking host calls, device calls, K5, a file decode): every result equals the
    blocking call's on a second context (itself held to the oracle by tests/test_gpu_decode.py), and the first case of every
    geometry the oracle's directly."""
    import torch
    import video_coding_amd as hvc
    from conftest import golden_bytes
    rng = np.random.Generator(np.random.PCG64(20261005))
    ref = hvc.Context(0)
    S = hvc.hvc.HVC_SLOTS
    geoms = [[(3, 2, 0)], [(9, 7, 0), (5, 4, 1), (5, 4, 1)], [(30, 17, 0), (15, 9, 1), (15, 9, 1)], [(1, 1, 0), (1, 1, 1)]]
    sets = []
    for gi, planes in enumerate(geoms):
        n = int(rng.integers(1, 6))
        specs, cfs, pfs, q, coefs, want, pix = _frame_set(n, planes, 7000 + 100 * gi)
        sets.append(dict(n=n, specs=specs, comps=hvc.hvc.components(specs), cfs=cfs, pfs=pfs, q=q, coefs=coefs, want=want, pix=pix))
    pinned = []
    in_flight = {}   # slot -> (kind, set, destination, keepalive)
    mini = golden_bytes("mini.jpg")
    mini_want = ref.jpeg_decode(mini)[1]

    def retire(slot):
        kind, st, dst, _ = in_flight.pop(slot)
        ctx.wait(slot)
        got = dst.cpu().numpy() if hasattr(dst, "cpu") else dst
        assert np.array_equal(got, st["want"] if kind == "dec" else st["coefs"]), (kind, slot)

    try:
        for it in range(120):
            slot = int(rng.integers(0, S))
            if slot in in_flight:
                retire(slot)
            st = sets[int(rng.integers(0, len(sets)))]
            kind = "dec" if rng.random() < 0.7 else "enc"
            src = st["coefs"] if kind == "dec" else st["pix"]
            if rng.random() < 0.5:   # a pinned copy of the source
                p = ctx.host_alloc(src.shape, src.dtype)
                p[:] = src
                pinned.append(p)
                src = p
            shape, dt = ((st["n"], st["pfs"]), np.uint8) if kind == "dec" else ((st["n"], st["cfs"]), np.int16)
            if rng.random() < 0.4:
                dst = torch.zeros(shape, dtype=torch.uint8 if kind == "dec" else torch.int16, device="cuda")
                torch.cuda.synchronize()
            else:
                dst = np.zeros(shape, dtype=dt)
            if kind == "dec":
                ctx.decode_frames_submit(slot, src, st["cfs"], st["q"], st["comps"], st["n"], dst, st["pfs"])
            else:
                ctx.encode_frames_submit(slot, src, st["pfs"], st["q"], st["comps"], st["n"], dst, st["cfs"])
            in_flight[slot] = (kind, st, dst, src)
            # something else on the same context while the slots are in flight
            what = int(rng.integers(0, 5))
            o = sets[int(rng.integers(0, len(sets)))]
            if what == 0:      # a blocking host-memory decode
                out = np.zeros((o["n"], o["pfs"]), dtype=np.uint8)
                ctx.decode_frames(o["coefs"], o["cfs"], o["q"], o["comps"], o["n"], out, o["pfs"])
                assert np.array_equal(out, o["want"])
            elif what == 1:    # a device-memory encode + K5 of its result against the host array's
                d_p = torch.from_numpy(o["pix"]).cuda()
                d_c = torch.zeros((o["n"], o["cfs"]), dtype=torch.int16, device="cuda")
                torch.cuda.synchronize()
                ctx.encode_frames(d_p, o["pfs"], o["q"], o["comps"], o["n"], d_c, o["cfs"])
                assert np.array_equal(ctx.checksum_records(d_c, o["cfs"] * 2, o["n"]), ctx.checksum_records(o["coefs"], o["cfs"] * 2, o["n"]))
            elif what == 2:    # a whole file
                assert np.array_equal(ctx.jpeg_decode(mini)[1], mini_want)
            elif what == 3 and in_flight:   # an early retirement, any slot
                retire(list(in_flight)[int(rng.integers(0, len(in_flight)))])
        for slot in list(in_flight):
            retire(slot)
    finally:
        for slot in list(in_flight):
            ctx.wait(slot)
        for p in pinned:
            ctx.host_free(p)
        ref.close()


def test_every_block_through_the_fixup_list():
    """the exactness contract's worst case: adversarial records on which every block fails the packed kernel's guard -- the
    fix-up list holds every block of the launch (its fixed grid strides over it), the frames are the int64 kernel's"""
    import bench_configs as bc
    r = bc.config_fixup(bc.make_args(frames=16, steps=2, warmup=1))
    assert r["checksum"]["verified"] is True and r["wide_path_blocks"] == r["all_blocks"] == 16 * 48960, r
