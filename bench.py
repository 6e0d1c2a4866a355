#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG block-transform path.

Workload (BASELINE.json configs[1]): 1080p 4:2:0 baseline frames, synthetic
*valid* coefficient blocks (Huffman bypassed), resident in HBM; one "step" is
one pass of the decode hot path (dequantise -> inverse zig-zag -> Chen-Wang
IDCT -> clip/level shift -> plane store) over one batch of frames through the
C ABI (hvc_decode_frames).  Metric: Mpixel/s decoded (cropped 1920x1080 luma
pixels per frame).

    python bench.py --gpus N --steps K --warmup W

Setup (untimed, before the W warm-up steps): input generation on the GPU and eight launches of the
step itself (page touch + clock ramp).  N > 1 is launched by the driver with torch.distributed.run
(one rank per GPU);
the path shards as independent frame batches: no data-path collective, weak
scaling (per-GPU batch fixed).  The barrier / max-over-ranks reduction below is
timing closure only.

Inputs: seeded synthetic pixel frames (video-coding_amd/synth.py) pushed through
this library's OWN forward path (k_encode, quality 75) on the GPU, outside the
timed region -- so the coefficients are encoder-producible.  (tests/test_gpu_fullsize_properties.py
checks this very batch shape against the oracle and across four implementations.)

The JSON line also carries
  roofline      achieved algorithmic GB/s of the dominant kernel (k_decode_packed:
                192 B per 8x8 block = 128 B int16 coefficients read + 64 B pixels
                written) over its HIP-event-timed duration, against 8 TB/s HBM;
                `traffic` = the HBM bytes of the committed rocprofv3 PMC passes;
  cpu_baseline  the CPU oracle (oracle/hvc_oracle.c, the restated model path,
                scalar, 1 thread) timed on this host on a bounded sample of the
                same workload.  The oracle is the checker, timed as a baseline
                only: this leg is its only use here (parity is the job of tests/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 1920, 1080
# Decoder.init geometry (decoder.ml:304-345) for 4:2:0: rounded to 16 -> 1920x1088
PLANES = [(240, 136, 0), (120, 68, 1), (120, 68, 1)]  # (blocks_w, blocks_h, qtab)
BLOCKS_PER_FRAME = sum(bw * bh for bw, bh, _ in PLANES)  # 48960
ALGO_BYTES_PER_BLOCK = 192  # 128 B read + 64 B written
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def make_distinct_frames(ctx, hvc, n_distinct, seed):
    """Coefficient records of n_distinct synthetic frames via the library's own forward path
    (hvc_encode_frames on the GPU, Quant_tables.scale 75).  Returns (device int16 tensor
    [n_distinct, coef_count], qtabs uint16 [2, 64])."""
    import torch
    from video_coding_amd.synth import synth_frame_pixels
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(PLANES)
    pix = np.stack([synth_frame_pixels(seed + 16 * f, PLANES) for f in range(n_distinct)])
    d_pix = torch.from_numpy(pix).cuda()
    d_coefs = torch.zeros((n_distinct, cfs), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_frames(d_pix, pfs, qtabs, hvc.hvc.components(specs), n_distinct, d_coefs, cfs)
    ctx.synchronize()
    return d_coefs, qtabs


def _oracle_frame(orc, rec, qtabs):
    off = 0
    for bw, bh, qt in PLANES:
        n = bw * bh * 64
        orc.dequant_idct_recon(rec[off:off + n], qtabs[qt], bw, bh)
        off += n


def cpu_baseline(frames, qtabs, min_seconds=10.0):
    """Oracle block stage (scalar C, int64, one block at a time) on the same frames: (i) one thread
    -- the model's CPU path as restated; (ii) the same code frame-sharded over the host cores this
    process may use (ctypes releases the GIL), SURVEY.md 8(d)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc
    done, t0 = 0, time.perf_counter()
    while True:
        _oracle_frame(orc, frames[done % len(frames)], qtabs)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # a one-GPU box's CPU share is 16 cores
    per_thread = max(4, int(done / dt * min_seconds / 2))  # about min_seconds/2 of wall time

    def worker(k):
        for i in range(per_thread):
            _oracle_frame(orc, frames[(k + i) % len(frames)], qtabs)

    t1 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(worker, range(cores)))
    dtm = time.perf_counter() - t1
    return {"value": round(done * W * H / dt / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same 1080p 4:2:0 workload, %.1f s, oracle/hvc_oracle.c "
                      "orc_dequant_idct_recon, 1 thread" % (done, dt),
            "all_cores": {"value": round(cores * per_thread * W * H / dtm / 1e6, 3), "unit": "Mpixel/s", "cores": cores,
                          "sample": "%d frames, %.1f s, frame-sharded threads" % (cores * per_thread, dtm)}}


def measured_traffic(frames):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs of this very
    command and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot collect counters
    itself; None when the profile is absent or was taken at another batch size."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        return round(t["hbm_bytes"]) if t.get("frames_per_launch") == frames else None
    except (OSError, ValueError, KeyError):
        return None


def dist_env():
    """RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run exports them (1 process = 1 GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def dist_init(world, backend, device=None):
    """Process group for timing closure only (barrier + MAX of the elapsed time): the data path
    has no collective.  backend: "nccl" (= RCCL, one rank per GPU) or "gloo" (CPU rehearsal)."""
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return dist


def timed_steps(step, steps, warmup, sync, world, dist=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both sides."""
    for _ in range(warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    return time.perf_counter() - t0


def max_over_ranks(dt, world, dist=None, device="cpu"):
    if world == 1:
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_mpixels(world, frames_per_gpu, steps, dt):
    """value = units ALL ranks processed / max-over-ranks time (weak scaling: per-GPU batch fixed)."""
    return world * frames_per_gpu * steps * W * H / dt / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # the first launches after idle run off-clock (DVFS)
    ap.add_argument("--frames", type=int, default=1024, help="frames per GPU per step (9.6 GB of coefficients + pixels)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames (replicated to --frames)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import video_coding_amd as hvc

    rank, world, local_rank = dist_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # Rehearsal on a one-GPU box (never the measured configuration): HVC_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and closes the timing with gloo instead of RCCL (which refuses two ranks per GPU).
    rehearsal = os.environ.get("HVC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = dist_init(world, "gloo" if rehearsal else "nccl", torch.device("cuda", local_rank))

    ctx = hvc.Context(local_rank)  # raises without a gfx950 GPU: there is no CPU fallback
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_distinct, qtabs = make_distinct_frames(ctx, hvc, args.distinct, seed=0x4A504547 + 1000 * rank)
    specs, cfs, pfs = hvc.hvc.frame_layout(PLANES)
    reps = (args.frames + args.distinct - 1) // args.distinct
    d_coefs = d_distinct.repeat(reps, 1)[:args.frames].contiguous()
    d_pix = torch.zeros((args.frames, pfs), dtype=torch.uint8, device="cuda")
    comps = hvc.hvc.components(specs)
    torch.cuda.synchronize()
    ctx.set_profiling(True)

    def step():
        ctx.decode_frames(d_coefs, cfs, qtabs, comps, args.frames, d_pix, pfs)

    # Setup, not measurement: a few launches so that the output buffer's pages are touched and the chip
    # is on its sustained clock even when the caller asks for very few warm-up steps (the first launches
    # after idle run off the sustained rate in either direction).
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    ctx.set_profiling(True)  # restart the event ring: only warm-up and timed steps are recorded from here

    dt = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world, dist)
    dt = max_over_ranks(dt, world, dist, "cpu" if rehearsal else "cuda")

    # HIP events recorded around k_decode_packed inside the timed region (one pair per step)
    kernel_ms = ctx.kernel_ms_history(min(args.steps, 64))
    wide = ctx.last_wide_blocks()
    frames_host = d_distinct.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        algo_bytes = args.frames * BLOCKS_PER_FRAME * ALGO_BYTES_PER_BLOCK
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "Mpixel/s decoded (1080p 4:2:0 batch)",
            "value": round(whole_job_mpixels(world, args.frames, args.steps, dt), 1),
            "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic" + (" (REHEARSAL: all ranks share cuda:0, gloo)" if rehearsal else ""),
            "config": {"workload": "1080p 4:2:0 baseline, synthetic valid coefficient blocks (Huffman bypassed), "
                                   "HBM-resident, %d frames/GPU/step" % args.frames,
                       "frames_per_gpu_per_step": args.frames, "blocks_per_frame": BLOCKS_PER_FRAME,
                       "parallelism": "independent frame batch per GPU, no collective",
                       "wide_path_blocks": int(wide)},
            "roofline": {"bound": "hbm", "kernel": "k_decode_packed", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": measured_traffic(args.frames), "kernel_ms": round(k_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "arithmetic": "int32 with int16-pair dot products (v_dot2_i32_i16); int64 fix-up kernel "
                                       "for blocks outside the proven range"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_host, qtabs, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
