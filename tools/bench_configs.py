#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (bench.py stays the headline contract):

  --config 3   batch of 1080p 4:2:0 baseline JPEGs: host Huffman (T threads) -> pinned ring ->
               hipMemcpyAsync on a copy stream || block-stage kernel (hvc_jpeg_decode_batch)
  --config 4   4K 4:4:4 decode, one GPU's shard shape (388 800 blocks/frame), HBM-resident
  --config 5   encoder: forward 8x8 DCT + quantise, 4K 4:2:0, HBM-resident (hvc_encode_frames)
  --config 2 / 10 / 7 / 8 / 9 / 6   K2 upsample / subsample_hv2 / fused 4:4:4 / config 5 to files / GPU Huffman coder / host buffers
  --config 11  whole `oyuv convert` passes (hvc_yuv_convert) on 1080p frames
  --config 12  the asynchronous seam: caller-filled pinned slots -> hvc_decode_frames_submit / hvc_wait (--host-out: pixels back too)
  --config 13  every block through the int64 kernel (--wide-mode kernel2 | dqt16)
  --config 14  every block through the fix-up list (adversarial coefficients under 8-bit tables)

Every config function returns its JSON object (bench.py collects them as `others` in its one line); run as a command,
this file prints it.

Each prints one JSON line.  Inputs are synthetic and are prepared with the library's own paths
(hvc_jpeg_encode / hvc_encode_frames) outside every timed region; nothing here touches the oracle.
What a configuration produced is checksummed on the device after the timed region (K5, hvc_checksum_records;
EVERY output record, not a sample) and compared with tests/golden/bench_checksums.json -- values the CPU suite
derives from the model restatement on the same seeds (tests/test_bench_checksums.py): `verified` in the line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def verify(ctx, data, record_bytes, n_records, key, distinct):
    """K5 over all n_records output records (record r holds distinct frame r % distinct) against the golden values"""
    sums = ["%016x" % int(x) for x in ctx.checksum_records(data, record_bytes, n_records)]
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json")) as f:
            want = json.load(f)[key]
    except (OSError, ValueError, KeyError):
        want = None
    ok = None
    if want is not None and distinct <= len(want):
        ok = all(sums[r] == want[r % distinct] for r in range(n_records))
    return {"checksum": {"records": n_records, "distinct": sums[:distinct], "expected": "tests/golden/bench_checksums.json:" + key,
                         "verified": ok}}


def config3_files(ctx, distinct):
    """configuration 3's input: `distinct` seeded 1080p 4:2:0 frames as baseline JPEG files (q75), written by the library's own
    encoder (hvc_jpeg_encode) -- tests/golden/bench_checksums.json `configs_c3` holds the model's decode of these"""
    from video_coding_amd.synth import synth_pixels
    W, H = 1920, 1080
    jpegs = []
    for f in range(distinct):
        y = synth_pixels(10 + f, 1088, 1920)[:H]
        u = synth_pixels(20 + f, 544, 960)[:H // 2]
        v = synth_pixels(30 + f, 544, 960)[:H // 2]
        jpegs.append(ctx.jpeg_encode(y, u, v, W, H, 420, 75))
    return jpegs


def config3(args):
    """-> the JSON object of configuration 3 (host or GPU reader)"""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_pixels
    W, H = 1920, 1080
    ctx = hvc.Context(0)
    jpegs = config3_files(ctx, args.distinct)
    ri = int(getattr(args, "restart_interval", 0) or 0)
    if ri:
        # every file re-written with a restart interval of ri MCUs (DRI + RSTn; 120 = a row of MCUs, what encoders write) and the
        # readers told to honour them (hvc_set_restart_markers: the opt-in extension) -- same coefficients, same frames
        from jpeg_opt_writer import jpeg_optimised_tables
        qt = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
        jpegs = [jpeg_optimised_tables(W, H, 420, qt, hvc.hvc.jpeg_entropy_decode(j)[1], restart_interval=ri) for j in jpegs]
        ctx.set_restart_markers(True)
    elif getattr(args, "own_tables", False):
        # every distinct file re-written with Huffman tables optimised for its own statistics (what libjpeg -optimize
        # writes): other tables than the model's defaults AND other tables from file to file -- same coefficients,
        # same decoded frames, so the golden checksums still apply
        from jpeg_opt_writer import jpeg_optimised_tables
        qt = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
        jpegs = [jpeg_optimised_tables(W, H, 420, qt, hvc.hvc.jpeg_entropy_decode(j)[1]) for j in jpegs]
    batch = [jpegs[i % len(jpegs)] for i in range(args.frames)]
    info = hvc.hvc.jpeg_read_header(batch[0])
    if args.host_out:   # decoded frames back in (pageable) host memory: the other PCIe direction joins in
        d_pix = np.zeros(args.frames * info.pixel_bytes, dtype=np.uint8)
    else:
        d_pix = torch.zeros(args.frames * info.pixel_bytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu = bool(getattr(args, "gpu_entropy", False))
    ctx.jpeg_decode_batch(batch[:min(64, args.frames)], d_pix, info.pixel_bytes, threads=args.threads,
                          frames_per_chunk=args.chunk, gpu_entropy=gpu)  # warm-up: allocates the pinned ring
    best = None
    for _ in range(args.steps):
        t0, c0 = time.perf_counter(), time.process_time()
        st = ctx.jpeg_decode_batch(batch, d_pix, info.pixel_bytes, threads=args.threads, frames_per_chunk=args.chunk,
                                   gpu_entropy=gpu)
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
        if best is None or dt < best[0]:
            best = (dt, st, cpu)
    dt, st, cpu = best
    jpeg_bytes = sum(len(j) for j in batch)
    result = {
        **verify(ctx, d_pix, info.pixel_bytes, args.frames, "configs_c3", args.distinct),
        "config": ("3-gpu-entropy" if gpu else "3") + ("-host-out" if args.host_out else "") +
                  ("-own-tables" if getattr(args, "own_tables", False) else "") + ("-restart-%d" % ri if ri else ""),
        "metric": "Mpixel/s decoded, " + ("host unstuffing + H2D of segments + GPU Huffman + GPU block stage"
                                          if gpu else "host Huffman + H2D + GPU block stage") + " overlapped",
        "host_prep_thread_ms_sum": round(st.host_prep_ms_sum, 1),
        "process_cpus_busy": round(cpu / dt, 2),  # CPU time of the whole process / wall time of the call
        "value": round(args.frames * W * H / dt / 1e6, 1), "unit": "Mpixel/s", "frames": args.frames,
        "host_threads": args.threads, "frames_per_chunk": st.frames_per_chunk, "chunks": st.chunks,
        "wall_ms": round(dt * 1e3, 2), "api_wall_ms": round(st.wall_ms, 2), "jpeg_MB": round(jpeg_bytes / 1e6, 1),
        "entropy_thread_ms_sum": round(st.entropy_ms_sum, 1),
        "entropy_Mpixel_s_per_thread": round(args.frames * W * H / (max(st.entropy_ms_sum, 1e-9) * 1e-3) / 1e6, 1),
        "h2d_ms_sum": round(st.h2d_ms_sum, 2), "h2d_GBps": round(st.coef_bytes / (max(st.h2d_ms_sum, 1e-9) * 1e-3) / 1e9, 1),
        "kernel_ms_sum": round(st.kernel_ms_sum, 2),
        "overlap": "sum of stage times / wall = %.2f" % ((st.entropy_ms_sum / args.threads + st.h2d_ms_sum +
                                                          st.kernel_ms_sum) / (dt * 1e3)),
        # the share of the stages' time the pipeline hides
        "overlap_fraction": round(1.0 - dt * 1e3 / max(st.entropy_ms_sum / args.threads + st.h2d_ms_sum + st.kernel_ms_sum, 1e-9), 3),
        "bound": ("GPU reader kernels / upload" if gpu else "host Huffman (entropy time / threads ~ wall)") +
                 (", download of the frames" if args.host_out else "")}
    ctx.close()
    return result


def resident_decode(args, planes, W, H, tag):
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
    # the resident batch as bench.py lays it out: planes and frames on 64 KiB / 2 MiB boundaries (hvc.layout_alignment)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes, align=1 if getattr(args, "tight", False) else "auto")
    comps = hvc.hvc.components(specs)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    # valid coefficients from the library's own forward path
    src = torch.from_numpy(np.stack([synth_frame_pixels(40 + 8 * f, planes) for f in range(args.distinct)])).cuda()
    d_distinct = torch.zeros((args.distinct, tcfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, tpfs, qtabs, hvc.hvc.components(tspecs), args.distinct, d_distinct, tcfs)
    reps = (args.frames + args.distinct - 1) // args.distinct
    d_coefs = hvc.hvc.spread_records(d_distinct, tspecs, specs, cfs, "coef_offset").repeat(reps, 1)[:args.frames].contiguous()
    d_pix = torch.zeros((args.frames, pfs), dtype=torch.uint8, device="cuda")
    ctx.set_profiling(True)
    for _ in range(args.warmup):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, args.frames, d_pix, pfs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, args.frames, d_pix, pfs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ctx.kernel_ms_history(min(args.steps, 64))))
    blocks = sum(bw * bh for bw, bh, _ in planes)
    algo = args.frames * blocks * 192
    result = {**verify(ctx, hvc.hvc.tight_records(d_pix, specs, "plane_offset"), tpfs, args.frames, "configs_c%d" % tag, args.distinct),
                      "config": tag, "metric": "Mpixel/s decoded", "value": round(args.frames * args.steps * W * H / dt / 1e6, 1),
                      "unit": "Mpixel/s", "frames": args.frames, "blocks_per_frame": blocks, "kernel_ms": round(k_ms, 4),
                      "algorithmic_GBps": round(algo / (k_ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(algo / (k_ms * 1e-3) / 8e12, 4),
                      "wide_path_blocks": int(ctx.last_wide_blocks())}
    ctx.close()
    return result


def config5(args):
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    W, H = 3840, 2160
    planes = [(480, 270, 0), (240, 135, 1), (240, 135, 1)]
    ql, qc = hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)
    tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes, align=1 if getattr(args, "tight", False) else "auto")   # (as bench.py --config 5)
    recs = [synth_frame_pixels(60 + 8 * f, planes) for f in range(args.distinct)]
    reps = (args.frames + args.distinct - 1) // args.distinct
    d_pix = hvc.hvc.spread_records(torch.from_numpy(np.stack(recs)).cuda(), tspecs, specs, pfs, "plane_offset").repeat(reps, 1)[:args.frames].contiguous()
    d_coefs = torch.zeros((args.frames, cfs), dtype=torch.int16, device="cuda")
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_profiling(True)
    comps = hvc.hvc.components(specs)
    qtabs = np.stack([ql, qc])
    for _ in range(args.warmup):
        ctx.encode_frames(d_pix, pfs, qtabs, comps, args.frames, d_coefs, cfs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.encode_frames(d_pix, pfs, qtabs, comps, args.frames, d_coefs, cfs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k_ms = float(np.mean(ctx.kernel_ms_history(min(args.steps, 64))))
    blocks = sum(bw * bh for bw, bh, _ in planes)
    algo = args.frames * blocks * 192
    result = {**verify(ctx, hvc.hvc.tight_records(d_coefs, specs, "coef_offset"), tcfs * 2, args.frames, "configs_c5", args.distinct),
                      "config": 5, "metric": "Mpixel/s encoded (fDCT + quantise, 4K 4:2:0)",
                      "value": round(args.frames * args.steps * W * H / dt / 1e6, 1), "unit": "Mpixel/s",
                      "frames": args.frames, "blocks_per_frame": blocks, "kernel_ms": round(k_ms, 4),
                      "algorithmic_GBps": round(algo / (k_ms * 1e-3) / 1e9, 1),
                      "frac_of_8TBps": round(algo / (k_ms * 1e-3) / 8e12, 4)}
    ctx.close()
    return result


def config_host(args):
    """The HVC_MEM_HOST form of the boundary (what a caller holding OCaml Bigarrays gets): pageable
    host coefficient records in, host pixel records out -- PCIe-inclusive, never the headline value."""
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    ctx = hvc.Context(0)
    src = np.stack([synth_frame_pixels(70 + 8 * f, planes) for f in range(args.distinct)])
    distinct = np.zeros((args.distinct, cfs), dtype=np.int16)
    ctx.encode_frames(src, pfs, qtabs, comps, args.distinct, distinct, cfs)  # host in, host out
    coefs = np.ascontiguousarray(np.tile(distinct, ((args.frames + args.distinct - 1) // args.distinct, 1))[:args.frames])
    pixels = np.zeros((args.frames, pfs), dtype=np.uint8)
    for _ in range(2):
        ctx.decode_frames(coefs, cfs, qtabs, comps, args.frames, pixels, pfs)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.decode_frames(coefs, cfs, qtabs, comps, args.frames, pixels, pfs)
    dt = (time.perf_counter() - t0) / args.steps
    moved = args.frames * (cfs * 2 + pfs)
    result = {"config": "host", "metric": "Mpixel/s decoded, host buffers in and out (PCIe-inclusive)",
                      "value": round(args.frames * 1920 * 1080 / dt / 1e6, 1), "unit": "Mpixel/s", "frames": args.frames,
                      "ms_per_call": round(dt * 1e3, 2), "bytes_over_pcie": moved,
                      "pcie_GBps": round(moved / dt / 1e9, 1)}
    ctx.close()
    return result


def config_async(args):
    """The asynchronous seam (include/hvc_jpeg.h: hvc_host_alloc, hvc_decode_frames_submit, hvc_wait): 1080p 4:2:0 coefficient
    records handed over in PINNED slot buffers, HVC_SLOTS batches in flight.  The caller keeps its own entropy reader (the model's,
    decoder.ml:118-140): here a pool of caller threads REFILLS slot k + 1's pinned buffer -- a copy of the next batch's records,
    which is what a reader's output amounts to for the link -- while the GPU works on slot k.  Records: the library's host reader
    on config 3's files (hvc_jpeg_entropy_decode), so the decoded frames are configs_c3's.
    args.host_out: the pixel records come back into pinned host buffers as well (and every one is compared, byte for byte, with
    the device-decoded frame of its distinct record); otherwise they stay in HBM and K5 verifies all of them.
    Reports Gpixel/s over the whole loop (refills included), the upload rate, and the overlap fraction
    1 - wall / (refill + upload + kernel + download), each summed over the submissions."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    import video_coding_amd as hvc
    W, H = 1920, 1080
    ctx = hvc.Context(0)
    jpegs = config3_files(ctx, args.distinct)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    cfs, pfs = info.coef_count, info.pixel_bytes
    planes = [(info.layout[i].blocks_w, info.layout[i].blocks_h, info.layout[i].qtab) for i in range(info.n_comp)]
    specs, cfs2, pfs2 = hvc.hvc.frame_layout(planes)
    assert (cfs2, pfs2) == (cfs, pfs)
    comps = hvc.hvc.components(specs)
    qtabs = info.qtab_array()
    distinct = np.stack([hvc.hvc.jpeg_entropy_decode(j)[1] for j in jpegs])        # [D][cfs] int16, pageable
    n, C, S, D = args.frames, args.chunk, hvc.hvc.HVC_SLOTS, args.distinct
    n_chunks = (n + C - 1) // C
    host_out = bool(getattr(args, "host_out", False))
    bufs = [ctx.host_alloc((C, cfs), np.int16) for _ in range(S)]
    outs = [ctx.host_alloc((C, pfs), np.uint8) for _ in range(S)] if host_out else None
    d_pix = None if host_out else torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    expect = None
    if host_out:   # the distinct frames decoded once through the blocking entry point (K5-verified below): what every record must equal
        expect = np.zeros((D, pfs), dtype=np.uint8)
        ctx.decode_frames(distinct, cfs, qtabs, comps, D, expect, pfs)
    pool = ThreadPoolExecutor(args.threads)
    T = max(1, args.threads)

    def fill_part(s, first, cnt, t):     # thread t's share of slot s's refill
        for f in range(t, cnt, T):
            np.copyto(bufs[s][f], distinct[(first + f) % D])
        return True

    def check_part(s, first, cnt, t):
        return all(np.array_equal(outs[s][f], expect[(first + f) % D]) for f in range(t, cnt, T))

    def one_pass():
        st = dict(fill=0.0, h2d=0.0, k=0.0, d2h=0.0, h2d_bytes=0, d2h_bytes=0, ok=True)
        where = [None] * S

        def retire(s):
            ctx.wait(s)
            x = ctx.slot_last_stats(s)
            st["h2d"] += x.h2d_ms
            st["k"] += x.kernel_ms
            st["d2h"] += x.d2h_ms
            st["h2d_bytes"] += x.h2d_bytes
            st["d2h_bytes"] += x.d2h_bytes

        t0 = time.perf_counter()
        for k in range(n_chunks):
            s, first = k % S, k * C
            cnt = min(C, n - first)
            checks = []
            if where[s] is not None:
                retire(s)
                if host_out:   # the consumer of slot s's pixels runs beside the refill of its coefficient buffer
                    pf, pc = where[s]
                    checks = [pool.submit(check_part, s, pf, pc, t) for t in range(T)]
            f0 = time.perf_counter()
            for fu in [pool.submit(fill_part, s, first, cnt, t) for t in range(T)]:
                fu.result()
            st["fill"] += (time.perf_counter() - f0) * 1e3
            st["ok"] &= all(c.result() for c in checks)
            ctx.decode_frames_submit(s, bufs[s], cfs, qtabs, comps, cnt, outs[s] if host_out else d_pix[first:first + cnt], pfs)
            where[s] = (first, cnt)
        for k in range(n_chunks, n_chunks + S):   # drain in submission order
            s = k % S
            if where[s] is not None:
                retire(s)
                if host_out:
                    pf, pc = where[s]
                    st["ok"] &= all(c.result() for c in [pool.submit(check_part, s, pf, pc, t) for t in range(T)])
                where[s] = None
        st["wall"] = (time.perf_counter() - t0) * 1e3
        return st

    one_pass()   # warm-up: the slots' device buffers, the streams, the clock
    best = None
    for _ in range(args.steps):
        st = one_pass()
        if best is None or st["wall"] < best["wall"]:
            best = st
    st = best
    if host_out:
        v = verify(ctx, expect, pfs, D, "configs_c3", D)
        v["checksum"]["records"] = n
        v["checksum"]["verified"] = bool(v["checksum"]["verified"]) and bool(st["ok"])
        v["checksum"]["how"] = "every downloaded record compared byte for byte with its distinct frame's, those K5-verified"
    else:
        v = verify(ctx, d_pix, pfs, n, "configs_c3", D)
    stages = st["fill"] + st["h2d"] + st["k"] + st["d2h"]
    result = {**v, "config": "async-seam" + ("-host-out" if host_out else ""),
              "metric": "Mpixel/s decoded: caller-filled pinned coefficient slots -> hvc_decode_frames_submit / hvc_wait" +
                        (" -> pinned pixel slots" if host_out else " -> HBM"),
              "value": round(n * W * H / (st["wall"] * 1e-3) / 1e6, 1), "unit": "Mpixel/s", "frames": n, "frames_per_slot": C,
              "slots": S, "host_threads": args.threads, "frames_per_chunk": C, "wall_ms": round(st["wall"], 2),
              "refill_ms_sum": round(st["fill"], 2), "h2d_ms_sum": round(st["h2d"], 2), "kernel_ms_sum": round(st["k"], 2),
              "d2h_ms_sum": round(st["d2h"], 2),
              "h2d_GBps": round(st["h2d_bytes"] / (max(st["h2d"], 1e-9) * 1e-3) / 1e9, 1),
              "d2h_GBps": round(st["d2h_bytes"] / (st["d2h"] * 1e-3) / 1e9, 1) if st["d2h"] > 0 else None,
              "refill_GBps": round(st["h2d_bytes"] / (max(st["fill"], 1e-9) * 1e-3) / 1e9, 1),
              "link_GBps_over_wall": round((st["h2d_bytes"] + st["d2h_bytes"]) / (st["wall"] * 1e-3) / 1e9, 1),
              "overlap_fraction": round(1.0 - st["wall"] / max(stages, 1e-9), 3),
              "bound": "pcie: 3.02 B of coefficients per pixel up" + (", 1.5 B of pixels down" if host_out else "")}
    pool.shutdown()
    for b in bufs + (outs or []):
        ctx.host_free(b)
    ctx.close()
    return result


def config_wide(args):
    """k_decode_wide_all, the model's 63-bit arithmetic for every block of a call: what a DQT with an entry above 255 costs (mode
    "dqt16": the luma table's first AC entry raised to 256, the coefficient there zeroed so that the frames -- and the golden
    checksums -- stay config 2's) and what hvc_set_decode_kernel(ctx, 2) costs (mode "kernel2").  1080p 4:2:0, HBM-resident."""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    W, H = 1920, 1080
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.from_numpy(np.stack([synth_frame_pixels(40 + 8 * f, planes) for f in range(args.distinct)])).cuda()
    d_distinct = torch.zeros((args.distinct, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, pfs, qtabs, comps, args.distinct, d_distinct, cfs)
    n = args.frames
    d_coefs = d_distinct.repeat((n + args.distinct - 1) // args.distinct, 1)[:n].contiguous()
    d_pix = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    d_ref = torch.zeros((args.distinct, pfs), dtype=torch.uint8, device="cuda")
    mode = getattr(args, "wide_mode", "kernel2")
    q_run = qtabs.copy()
    if mode == "dqt16":
        # a 16-bit table: entry 63 (the last zig-zag position) of both tables becomes 40000 and the coefficient it scales is
        # zeroed in every record, so the frames are the 8-bit tables' frames of the same (modified) records -- which the packed
        # kernel decodes below as the reference.  (16-bit entries that DO scale something: tests/test_gpu_decode.py, oracle.)
        q_run[:, 63] = 40000
        d_coefs.view(n, -1, 64)[:, :, 63] = 0
        d_distinct.view(args.distinct, -1, 64)[:, :, 63] = 0
    else:
        ctx.set_decode_kernel(2)
    # reference: the same records through the default (packed) kernel with the 8-bit tables
    ref = hvc.Context(0)
    ref.set_stream(torch.cuda.current_stream().cuda_stream)
    ref.decode_frames(d_distinct, cfs, qtabs, comps, args.distinct, d_ref, pfs)
    ref.synchronize()
    ref.close()
    ctx.set_profiling(True)
    for _ in range(args.warmup):
        ctx.decode_frames(d_coefs, cfs, q_run, comps, n, d_pix, pfs)
    torch.cuda.synchronize()
    for _ in range(args.steps):
        ctx.decode_frames(d_coefs, cfs, q_run, comps, n, d_pix, pfs)
    torch.cuda.synchronize()
    k_ms = float(np.mean(ctx.kernel_ms_history(min(args.steps, 64))))
    wide = int(ctx.last_wide_blocks())
    same = bool((d_pix.view(-1, args.distinct, pfs) == d_ref[None]).all()) if n % args.distinct == 0 else None
    blocks = sum(bw * bh for bw, bh, _ in planes)
    algo = n * blocks * 192
    result = {"config": "wide-" + mode, "metric": "Mpixel/s decoded, every block through the int64 kernel", "frames": n,
              "value": round(n * W * H / (k_ms * 1e-3) / 1e6, 1), "unit": "Mpixel/s", "kernel_ms": round(k_ms, 4),
              "algorithmic_GBps": round(algo / (k_ms * 1e-3) / 1e9, 1), "frac_of_8TBps": round(algo / (k_ms * 1e-3) / 8e12, 4),
              "wide_path_blocks": wide, "all_blocks": n * blocks,
              "checksum": {"records": n, "verified": same and wide == n * blocks,
                           "how": "every record equal to the packed kernel's decode of the same distinct record" +
                                  (" (coefficient 63 zeroed: the 16-bit entry scales nothing)" if mode == "dqt16" else "")}}
    ctx.close()
    return result


def config_fixup(args):
    """The worst case of the exactness contract with 8-bit tables: adversarial records (every coefficient a random value in
    +-2047) on which EVERY block fails k_decode_packed's guard, so the default path = the packed kernel (whose output is
    discarded block by block) + the fix-up list + k_decode_wide over the whole list.  Timed as a whole call (hvc_timer:
    packed + fix-up kernels); every record equal to k_decode_wide_all's frames of the same records (hvc_set_decode_kernel 2)."""
    import torch
    import video_coding_amd as hvc
    W, H = 1920, 1080
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    n, D = args.frames, args.distinct
    rng = np.random.Generator(np.random.PCG64(4242))
    d_distinct = torch.from_numpy(rng.integers(-2047, 2048, size=(D, cfs)).astype(np.int16)).cuda()
    d_coefs = d_distinct.repeat((n + D - 1) // D, 1)[:n].contiguous()
    d_pix = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    d_ref = torch.zeros((D, pfs), dtype=torch.uint8, device="cuda")
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_decode_kernel(2)
    ctx.decode_frames(d_distinct, cfs, qtabs, comps, D, d_ref, pfs)
    ctx.synchronize()
    ctx.set_decode_kernel(0)
    for _ in range(args.warmup):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, n, d_pix, pfs)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(args.steps):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, n, d_pix, pfs)
    ms = ctx.timer_end() / args.steps
    wide = int(ctx.last_wide_blocks())
    ctx.set_profiling(True)   # the packed kernel alone (its event pair excludes the fix-up kernel): what the flagging costs it
    for _ in range(args.steps):
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, n, d_pix, pfs)
    packed_ms = float(np.mean(ctx.kernel_ms_history(args.steps)))
    blocks = sum(bw * bh for bw, bh, _ in planes)
    same = bool((d_pix.view(-1, D, pfs) == d_ref[None]).all()) if n % D == 0 else None
    result = {"config": "fixup-all-blocks", "metric": "Mpixel/s decoded, every block failing the packed kernel's guard (adversarial +-2047 coefficients)",
              "frames": n, "value": round(n * W * H / (ms * 1e-3) / 1e6, 1), "unit": "Mpixel/s", "ms_per_call": round(ms, 4),
              "packed_kernel_ms": round(packed_ms, 4), "fixup_kernel_ms": round(ms - packed_ms, 4),
              "frac_of_8TBps": round(n * blocks * 192 / (ms * 1e-3) / 8e12, 4), "wide_path_blocks": wide, "all_blocks": n * blocks,
              "checksum": {"records": n, "verified": same and wide == n * blocks,
                           "how": "every record equal to k_decode_wide_all's decode of the same distinct record"}}
    ctx.close()
    return result


def config_444(args):
    """next-3: 1080p 4:2:0 coefficient records -> tight 4:4:4 frames.  Fused (k_decode_444 + seam pass)
    against the three-launch composition hvc_decode_frames -> crop view -> hvc_upsample420 x 2."""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    W, H = 1920, 1080
    planes = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    comps = hvc.hvc.components(specs)
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.from_numpy(np.stack([synth_frame_pixels(90 + 8 * f, planes) for f in range(args.distinct)])).cuda()
    d_distinct = torch.zeros((args.distinct, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, pfs, qtabs, comps, args.distinct, d_distinct, cfs)
    n = args.frames
    d_coefs = d_distinct.repeat((n + args.distinct - 1) // args.distinct, 1)[:n].contiguous()
    d_out = torch.zeros((n, 3 * W * H), dtype=torch.uint8, device="cuda")
    d_pix = torch.zeros((n, pfs), dtype=torch.uint8, device="cuda")
    d_ref = torch.zeros((n, 3 * W * H), dtype=torch.uint8, device="cuda")

    def fused():
        ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out)

    def separate():
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, n, d_pix, pfs)
        for i in (1, 2):  # chroma: padded 960x544 plane, crop = the first 540 rows (a view), upsample
            ctx.upsample420(d_pix[:, specs[i]["plane_offset"]:], W // 2, H // 2, d_ref[:, i * W * H:], n_planes=n,
                            src_stride=960, dst_stride=W, src_plane_stride=pfs, dst_plane_stride=3 * W * H)

    res = {}
    for name, fn in (("fused", fused),) if getattr(args, "fused_only", False) else (("fused", fused), ("separate", separate)):
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        ctx.timer_begin()
        for _ in range(args.steps):
            fn()
        res[name] = ctx.timer_end() / args.steps
    blocks_needed = 240 * 135 + 2 * 120 * 68
    algo = n * (blocks_needed * 128 + 3 * W * H)
    ctx.decode_frames_yuv444(d_coefs, cfs, qtabs, comps, n, W, H, d_out)  # (the separate path ran last: the fused output once more)
    result = {**verify(ctx, d_out, 3 * W * H, n, "configs_c7", args.distinct),
                      "config": "444", "metric": "Mpixel/s decoded to 4:4:4 (1080p 4:2:0 in)", "frames": n,
                      "fused_ms": round(res["fused"], 4), "separate_ms": round(res["separate"], 4) if "separate" in res else None,
                      "value": round(n * W * H / (res["fused"] * 1e-3) / 1e6, 1), "unit": "Mpixel/s",
                      "speedup_vs_separate": round(res["separate"] / res["fused"], 3) if "separate" in res else None,
                      "algorithmic_GBps": round(algo / (res["fused"] * 1e-3) / 1e9, 1), "algorithmic_bytes": algo,
                      "frac_of_8TBps": round(algo / (res["fused"] * 1e-3) / 8e12, 4),
                      "wide_path_blocks": int(ctx.last_wide_blocks())}
    ctx.close()
    return result


def config5_files(args):
    """config 5 end to end: 4K 4:2:0 raw frames in, JPEG files out (hvc_jpeg_encode_batch):
    host pad (T threads) -> pinned ring -> H2D (copy stream) || k_encode + D2H || host RLE + Huffman (T threads)."""
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_pixels
    W, H = 3840, 2160
    ctx = hvc.Context(0)
    distinct = []
    for f in range(args.distinct):
        y = synth_pixels(110 + f, H, W)
        u = synth_pixels(120 + f, H // 2, W // 2)
        v = synth_pixels(130 + f, H // 2, W // 2)
        distinct.append(np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]))
    frames = [distinct[i % len(distinct)] for i in range(args.frames)]
    gpu = bool(getattr(args, "gpu_entropy", False))
    ctx.jpeg_encode_batch(frames[:min(2 * args.chunk, args.frames)], W, H, 420, 75, threads=args.threads,
                          frames_per_chunk=args.chunk, gpu_entropy=gpu)  # warm-up: allocates the pinned rings
    best = None
    for _ in range(args.steps):
        t0 = time.perf_counter()
        jpegs, st = ctx.jpeg_encode_batch(frames, W, H, 420, 75, threads=args.threads, frames_per_chunk=args.chunk,
                                          gpu_entropy=gpu)
        dt = time.perf_counter() - t0
        if best is None or st.wall_ms < best[1].wall_ms:
            best = (dt, st, jpegs)
    py_dt, st, jpegs = best
    # what came out: K5 over the bytes of the first `distinct` files (a file is one record) against the model encoder's
    # files for these seeds, and every later file equal to the one of its distinct frame
    sums = ["%016x" % int(ctx.checksum_records(np.frombuffer(jpegs[f], dtype=np.uint8), len(jpegs[f]), 1)[0]) for f in range(args.distinct)]
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_checksums.json")) as f:
            want = json.load(f)["configs_c5_files"]
    except (OSError, ValueError, KeyError):
        want = None
    ok = None
    if want is not None and args.distinct <= len(want):
        ok = sums == want[:args.distinct] and all(jpegs[f] == jpegs[f % args.distinct] for f in range(args.frames))
    checksum = {"checksum": {"records": args.frames, "distinct": sums, "expected": "tests/golden/bench_checksums.json:configs_c5_files",
                             "verified": ok}}
    # the call itself as a C caller sees it (hvc_batch_stats.wall_ms); the Python wrapper around it allocates 256 output
    # arrays and copies every file into a bytes object, which is not the library's time
    dt = st.wall_ms * 1e-3
    result = {
        **checksum,
        "config": "5-files" + ("-gpu-entropy" if gpu else ""), "python_wrapper_wall_ms": round(py_dt * 1e3, 2),
        "metric": "Mpixel/s encoded to JPEG files, host pad + H2D + GPU fDCT/quantise + " +
                  ("GPU Huffman + D2H of segments + host assembly" if gpu else "D2H + host Huffman") + " overlapped",
        "value": round(args.frames * W * H / dt / 1e6, 1), "unit": "Mpixel/s", "frames": args.frames,
        "host_threads": args.threads, "frames_per_chunk": st.frames_per_chunk, "chunks": st.chunks,
        "wall_ms": round(dt * 1e3, 2), "jpeg_MB": round(sum(len(j) for j in jpegs) / 1e6, 1),
        "pad_thread_ms_sum": round(st.host_prep_ms_sum, 1), "entropy_thread_ms_sum": round(st.entropy_ms_sum, 1),
        "entropy_Mpixel_s_per_thread": round(args.frames * W * H / (max(st.entropy_ms_sum, 1e-9) * 1e-3) / 1e6, 1),
        "h2d_ms_sum": round(st.h2d_ms_sum, 2), "kernel_ms_sum": round(st.kernel_ms_sum, 2),
        "d2h_ms_sum": round(st.d2h_ms_sum, 2), "d2h_GBps": round(st.coef_bytes / (st.d2h_ms_sum * 1e-3) / 1e9, 1)}
    ctx.close()
    return result


def config_huffman(args):
    """Encoder back end on the GPU: 4K 4:2:0 coefficient records (device resident, from k_encode) ->
    entropy-coded segments (hvc_huffman_encode_frames: length pass, scan, emit pass, stuffing passes)."""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    W, H = 3840, 2160
    info = hvc.hvc.jpeg_encoder_layout(W, H, 420, 75)
    planes = [(info.layout[i].blocks_w, info.layout[i].blocks_h, info.layout[i].qtab) for i in range(3)]
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    qtabs = info.qtab_array()[:2]
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    src = torch.from_numpy(np.stack([synth_frame_pixels(140 + 8 * f, planes) for f in range(args.distinct)])).cuda()
    d_distinct = torch.zeros((args.distinct, cfs), dtype=torch.int16, device="cuda")
    ctx.encode_frames(src, pfs, qtabs, hvc.hvc.components(specs), args.distinct, d_distinct, cfs)
    n = args.frames
    d_coefs = d_distinct.repeat((n + args.distinct - 1) // args.distinct, 1)[:n].contiguous()
    cap = n * 6 * 1024 * 1024
    out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    offs = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    import ctypes as C
    call = lambda: hvc.hvc._chk(hvc.lib().hvc_huffman_encode_frames(ctx._h, C.byref(info), d_coefs.data_ptr(), cfs, n,
                                                                    out.data_ptr(), cap, offs.data_ptr(), 1))
    for _ in range(args.warmup):
        call()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    dt = (time.perf_counter() - t0) / args.steps
    seg_bytes = int(offs[-1].item())
    result = {"config": "huffman", "metric": "Mpixel/s entropy-coded on the GPU (4K 4:2:0, q75)", "frames": n,
                      "value": round(n * W * H / dt / 1e6, 1), "unit": "Mpixel/s", "ms_per_call": round(dt * 1e3, 3),
                      "segment_MB": round(seg_bytes / 1e6, 1), "bits_per_pixel": round(seg_bytes * 8 / (n * W * H), 2),
                      "coef_GBps": round(n * cfs * 2 / dt / 1e9, 1)}
    ctx.close()
    return result


def _random_planes(seed, n_distinct, h, w):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=(n_distinct, h, w)).astype(np.uint8)


def config_k2(args):
    """K2: 4:2:0 -> 4:4:4 chroma upsample of 1080p chroma planes (960x540 -> 1920x1080), 2 planes/frame."""
    import torch
    import video_coding_amd as hvc
    cw, ch = 960, 540
    n = 2 * args.frames
    src = _random_planes(5, args.distinct, ch, cw)
    reps = (n + args.distinct - 1) // args.distinct
    d_src = torch.from_numpy(src).cuda().repeat(reps, 1, 1)[:n].contiguous()
    d_dst = torch.zeros((n, 2 * ch, 2 * cw), dtype=torch.uint8, device="cuda")
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(args.warmup):
        ctx.upsample420(d_src, cw, ch, d_dst, n_planes=n)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(args.steps):
        ctx.upsample420(d_src, cw, ch, d_dst, n_planes=n)
    ms = ctx.timer_end() / args.steps
    algo = n * cw * ch * 5  # 1 B read + 4 B written per source pixel
    result = {**verify(ctx, d_dst, 4 * cw * ch, n, "configs_k2", args.distinct),
              "config": "k2", "metric": "chroma samples/s upsampled 4:2:0 -> 4:4:4", "planes": n,
              "kernel_ms": round(ms, 4), "algorithmic_GBps": round(algo / (ms * 1e-3) / 1e9, 1),
              "frac_of_8TBps": round(algo / (ms * 1e-3) / 8e12, 4)}
    ctx.close()
    return result


def config_sub420(args):
    """Planar_444.subsample_hv2 (the chroma planes of `oyuv convert` 4:4:4 -> 4:2:0): 1920x1080 planes -> 960x540, 2 planes
    per frame; 5 B per destination sample (4 read + 1 written)."""
    import torch
    import video_coding_amd as hvc
    w, h = 1920, 1080
    n = 2 * args.frames
    src = _random_planes(6, args.distinct, h, w)
    reps = (n + args.distinct - 1) // args.distinct
    d_src = torch.from_numpy(src).cuda().repeat(reps, 1, 1)[:n].contiguous()
    d_dst = torch.zeros((n, h // 2, w // 2), dtype=torch.uint8, device="cuda")
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(args.warmup):
        ctx.subsample420(d_src, w, h, d_dst, n_planes=n)
    torch.cuda.synchronize()
    ctx.timer_begin()
    for _ in range(args.steps):
        ctx.subsample420(d_src, w, h, d_dst, n_planes=n)
    ms = ctx.timer_end() / args.steps
    algo = n * (w // 2) * (h // 2) * 5
    result = {**verify(ctx, d_dst, (w // 2) * (h // 2), n, "configs_sub420", args.distinct),
              "config": "sub420", "metric": "chroma samples/s subsampled 4:4:4 -> 4:2:0", "planes": n,
              "kernel_ms": round(ms, 4), "algorithmic_GBps": round(algo / (ms * 1e-3) / 1e9, 1),
              "frac_of_8TBps": round(algo / (ms * 1e-3) / 8e12, 4)}
    ctx.close()
    return result


# the passes of config 11: (key, input format, input size, output format, output size, offset) -- 1080p frames
CONVERT_PASSES = [("420_to_444", "420", (1920, 1080), "444", (1920, 1080), (0, 0)),
                  ("444_to_420", "444", (1920, 1080), "420", (1920, 1080), (0, 0)),
                  ("420_to_uyvy", "420", (1920, 1080), "UYVY", (1920, 1080), (0, 0)),
                  ("yuy2_to_420", "YUY2", (1920, 1080), "420", (1920, 1080), (0, 0)),
                  ("420_crop_720p", "420", (1920, 1080), "420", (1280, 720), (320, 180))]


def convert_input(key_index, n_distinct, frame_bytes):
    """the seeded raw frames of pass number key_index (tests/golden/make_bench_checksums.py makes the same ones)"""
    return np.random.Generator(np.random.PCG64(200 + key_index)).integers(0, 256, size=(n_distinct, frame_bytes)).astype(np.uint8)


def config_convert(args):
    """`oyuv convert` (Oconv.main, oconv.ml:111-133) on whole raw frames resident in HBM, hvc_yuv_convert: every pass's
    frames per second and its bytes in + bytes out per second (the passes go through a full-size 4:4:4 frame as the
    reference's do, so this rate is end to end, not a kernel's)."""
    import torch
    import video_coding_amd as hvc
    ctx = hvc.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    passes = []
    for i, (key, fi, si, fo, so, off) in enumerate(CONVERT_PASSES):
        fin, fout = hvc.YUV_FORMATS[fi], hvc.YUV_FORMATS[fo]
        in_fs, out_fs = hvc.yuv_frame_bytes(fin, *si), hvc.yuv_frame_bytes(fout, *so)
        src = convert_input(i, args.distinct, in_fs)
        reps = (args.frames + args.distinct - 1) // args.distinct
        d_src = torch.from_numpy(src).cuda().repeat(reps, 1)[:args.frames].contiguous()
        d_dst = torch.zeros((args.frames, out_fs), dtype=torch.uint8, device="cuda")
        for _ in range(args.warmup):
            ctx.yuv_convert(d_src, fin, si, d_dst, fout, so, off, n_frames=args.frames)
        torch.cuda.synchronize()
        ctx.timer_begin()
        for _ in range(args.steps):
            ctx.yuv_convert(d_src, fin, si, d_dst, fout, so, off, n_frames=args.frames)
        ms = ctx.timer_end() / args.steps
        v = verify(ctx, d_dst, out_fs, args.frames, "configs_convert_" + key, args.distinct)
        passes.append({"pass": key, "ms": round(ms, 4), "frames_per_s": round(args.frames / (ms * 1e-3), 1),
                       "in_plus_out_GBps": round(args.frames * (in_fs + out_fs) / (ms * 1e-3) / 1e9, 1),
                       "verified": v["checksum"]["verified"]})
        del d_src, d_dst
    ctx.close()
    oks = [q["verified"] for q in passes]
    return {"config": "convert", "metric": "oyuv convert passes, 1080p frames resident in HBM", "frames": args.frames,
            "passes": passes, "checksum": {"verified": None if None in oks else all(oks),
                                           "expected": "tests/golden/bench_checksums.json:configs_convert_*"}}


def make_args(**kw):
    """the argument object of the config functions for callers that are not this file's command line (bench.py)"""
    d = dict(frames=None, distinct=4, steps=None, warmup=10, threads=min(16, len(os.sched_getaffinity(0))), chunk=32,
             gpu_entropy=False, host_out=False, own_tables=False, fused_only=False, restart_interval=0, tight=False,
             wide_mode="kernel2")
    d.update(kw)
    return argparse.Namespace(**d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, required=True, choices=[2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
    ap.add_argument("--wide-mode", default="kernel2", choices=["kernel2", "dqt16"], help="config 13: what sends every block to the int64 kernel")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--distinct", type=int, default=4)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 16))
    ap.add_argument("--chunk", type=int, default=32)
    ap.add_argument("--gpu-entropy", action="store_true", help="configs 3 / 8: Huffman decoding / coding on the GPU as well")
    ap.add_argument("--host-out", action="store_true", help="config 3: decoded frames to host memory instead of HBM")
    ap.add_argument("--own-tables", action="store_true", help="config 3: every file with Huffman tables optimised for itself")
    ap.add_argument("--fused-only", action="store_true", help="config 7: skip the three-launch composition")
    ap.add_argument("--tight", action="store_true", help="configs 4 / 5: planes and frames back to back instead of on 64 KiB / 2 MiB boundaries (A/B)")
    ap.add_argument("--restart-interval", type=int, default=0, help="config 3: the files carry DRI / RSTn every so many MCUs (own tables too) and the readers honour them")
    args = ap.parse_args()
    if args.config == 12:  # the asynchronous seam: pinned slots, submit / wait
        args.frames = args.frames or 4096
        args.steps = args.steps or 2
        if args.chunk == 32:
            args.chunk = 64
        r = config_async(args)
    elif args.config == 14:  # every block through the fix-up list
        args.frames = args.frames or 64
        args.steps = args.steps or 5
        r = config_fixup(args)
    elif args.config == 13:  # every block through the int64 kernel
        args.frames = args.frames or 64
        args.steps = args.steps or 10
        r = config_wide(args)
    elif args.config == 2:  # K2 upsample (optional output stage)
        args.frames = args.frames or 256
        args.steps = args.steps or 20
        r = config_k2(args)
    elif args.config == 10:  # subsample_hv2 (oyuv convert 4:4:4 -> 4:2:0)
        args.frames = args.frames or 256
        args.steps = args.steps or 20
        r = config_sub420(args)
    elif args.config == 11:  # whole `oyuv convert` passes
        args.frames = args.frames or 256
        args.steps = args.steps or 10
        r = config_convert(args)
    elif args.config == 9:  # GPU Huffman coder alone
        args.frames = args.frames or 32
        args.steps = args.steps or 10
        r = config_huffman(args)
    elif args.config == 8:  # config 5 with files out
        args.frames = args.frames or 256
        args.steps = args.steps or 3
        if args.chunk == 32:
            args.chunk = 16
        r = config5_files(args)
    elif args.config == 7:  # fused 4:4:4 output (next-3)
        args.frames = args.frames or 512
        args.steps = args.steps or 20
        r = config_444(args)
    elif args.config == 6:  # host-buffer boundary of config 2
        args.frames = args.frames or 128
        args.steps = args.steps or 5
        r = config_host(args)
    elif args.config == 3:
        args.frames = args.frames or 256
        args.steps = args.steps or 3
        r = config3(args)
    elif args.config == 4:
        # one GPU's shard of config 4 = 2048 frames, processed as 16 resident chunks of 128 frames
        # (9.6 GB per launch), re-using the same device-resident synthetic chunk
        args.frames = args.frames or 128
        args.steps = args.steps or 16
        r = resident_decode(args, [(480, 270, 0), (480, 270, 1), (480, 270, 1)], 3840, 2160, 4)
    else:
        args.frames = args.frames or 256
        args.steps = args.steps or 30
        r = config5(args)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
