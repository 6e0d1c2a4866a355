#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG block-transform path.

    python bench.py --gpus N --steps K --warmup W [--config 2|3|4|5]

--config 2 (default; BASELINE.json configs[1], the configuration the metric is quoted on):
    1080p 4:2:0 baseline frames, synthetic *valid* coefficient blocks (Huffman bypassed), resident in
    HBM; one "step" = one pass of the decode hot path (dequantise -> inverse zig-zag -> Chen-Wang IDCT ->
    clip / level shift -> plane store) over one batch of --frames frames per GPU through the C ABI
    (hvc_decode_frames).  Metric: Mpixel/s decoded (cropped 1920x1080 luma pixels per frame).
--config 4 (BASELINE.json configs[3]: 16384 x 4K 4:4:4 sharded over 8 GPUs = 2048 frames per GPU):
    one "step" = one pass over the rank's whole shard, --shard frames in launches of --frames (128) frames;
    the shard is resident in HBM (102 GB of coefficients + 51 GB of pixels per GPU) when the device has the
    room, otherwise one resident launch-sized chunk is processed shard / frames times (said in `config`).

--config 3 (BASELINE.json configs[2]: 4096 x 1080p 4:2:0 baseline JPEG FILES, host Huffman || H2D || K1):
    the 4096 files are split over the ranks (4096 / N each: strong scaling); one "step" = one hvc_jpeg_decode_batch call over
    the rank's files -- host Huffman threads -> pinned ring -> hipMemcpyAsync on a side stream || k_decode_packed.  Each rank's
    host threads run on its share of the CPUs of its GPU's NUMA node (node CPUs / ranks on that node, 16 at most): this is the
    configuration where 8 ranks contend for the host.  Both readers are timed, K steps each: the host Huffman
    reader (`value`: what BASELINE words) and the GPU Huffman reader (`gpu_reader` in the line).  Bound: host, not HBM.
--config 5 (BASELINE.json configs[4]: encoder path, 4K 4:2:0 batch): one "step" = one hvc_encode_frames launch (level shift
    -> forward Chen DCT -> quantise -> zig-zag, k_encode) over --frames (256) HBM-resident frames per GPU; weak scaling.

N > 1: one process per GPU.  The driver starts the ranks with torch.distributed.run; typed by hand,
`python bench.py --gpus N` starts them itself -- N fresh child processes, before this process has made any
HIP or torch.cuda call -- and relays rank 0's JSON line.  The path shards as independent frame batches:
no data-path collective, weak scaling (per-GPU work fixed); the barrier / MAX-over-ranks reduction is timing
closure only.  HVC_BENCH_REHEARSAL=1 puts every rank on cuda:0 with gloo (a one-GPU box can rehearse the
whole N-rank flow; never a measurement, the line says so).

Setup (untimed, before the W warm-up steps): input generation on the GPU and a few launches of the step
itself (page touch + clock ramp).  Inputs: seeded synthetic pixel frames (video-coding_amd/synth.py) pushed
through this library's OWN forward path (k_encode, quality 75) on the GPU, outside the timed region -- so the
coefficients are encoder-producible.

The command exits non-zero when the decoded frames' checksums differ from the golden ones on any rank.

The JSON line also carries
  roofline      achieved algorithmic GB/s of the dominant kernel (k_decode_packed: 192 B per 8x8 block =
                128 B int16 coefficients read + 64 B pixels written) over its HIP-event-timed duration (events
                recorded inside the library on the kernel's own stream), against 8 TB/s HBM; `traffic` = the
                HBM bytes of the committed rocprofv3 PMC passes, `traffic_source` says which;
  checksum      K5 (hvc_checksum_records) over the decoded DISTINCT frames, on the device, compared with the
                values tests/golden/bench_checksums.json holds for these seeds -- which the CPU suite
                reproduces from the model restatement (tests/test_bench_checksums.py): the timed output is
                the model's output, not just fast;
                roofline.other_layout: the same frames with planes back to back (what the library's own entry points lay out;
                the headline batch starts every plane on a 64 KiB / 2 MiB boundary), timed and K5-verified after the timed region;
  sustained     >= --sustain-seconds (6 s) of the same step repeated AFTER the K timed steps, one event pair per step:
                first / last decile of the step time (clock or thermal drift would show), the GPU-busy fraction; never
                part of `value` / `ms_per_step`, which come from the K timed steps alone;
  others        (1 GPU, config 2) the other figures of DESIGN.md measured in this run, after the timed region: config 4's launch
                shape, K3, the fused 4:4:4 kernel, K2, subsample_hv2 (ms, fraction of 8 TB/s), configs 3 and 5 as pipelines
                with the host and the GPU entropy stage (Gpixel/s, threads, what binds them); each K5-verified;
  cpu_baseline  the CPU restatement of the model path (oracle/hvc_oracle.c, scalar, 1 thread) timed on this
                host on a bounded sample of the same workload.  The oracle is the checker, timed as a
                baseline only: this leg is its only use here (parity is the job of tests/).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_BLOCK = 192  # 128 B read + 64 B written
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec

# Decoder.init geometry (decoder.ml:304-345): (blocks_w, blocks_h, qtab) per component
WORKLOADS = {
    2: dict(name="1080p 4:2:0 baseline, synthetic valid coefficient blocks (Huffman bypassed), HBM-resident",
            W=1920, H=1080, planes=[(240, 136, 0), (120, 68, 1), (120, 68, 1)],  # rounded to 16 -> 1920x1088
            frames=1024, shard=None, steps=50, seed=0x4A504547, metric="Mpixel/s decoded (1080p 4:2:0 batch)"),
    4: dict(name="4K 4:4:4 baseline, synthetic valid coefficient blocks, one GPU's 2048-frame shard of the 16384-frame batch",
            W=3840, H=2160, planes=[(480, 270, 0), (480, 270, 1), (480, 270, 1)],
            frames=128, shard=2048, steps=3, seed=0x4A504547 + 400, metric="Mpixel/s decoded (4K 4:4:4 batch)"),
    3: dict(name="4096 x 1080p 4:2:0 baseline JPEG files, host Huffman || H2D || K1 (hvc_jpeg_decode_batch); the GPU Huffman reader beside it",
            W=1920, H=1080, planes=[(240, 136, 0), (120, 68, 1), (120, 68, 1)], frames=4096, shard=None, steps=2, seed=None,
            metric="Mpixel/s decoded (1080p 4:2:0 files, host Huffman + GPU block stage overlapped)"),
    5: dict(name="4K 4:2:0 encoder path: level shift -> forward Chen DCT -> quantise -> zig-zag (k_encode), HBM-resident frames",
            W=3840, H=2160, planes=[(480, 270, 0), (240, 135, 1), (240, 135, 1)], frames=256, shard=None, steps=40, seed=None,
            metric="Mpixel/s encoded (4K 4:2:0 batch, fDCT + quantise)"),
}
# kept for callers of the N > 1 helpers (tests/test_distributed_cpu.py)
W, H = WORKLOADS[2]["W"], WORKLOADS[2]["H"]
PLANES = WORKLOADS[2]["planes"]
BLOCKS_PER_FRAME = sum(bw * bh for bw, bh, _ in PLANES)  # 48960


def distinct_seed(base_seed, rank, f):
    return base_seed + 1000 * rank + 16 * f


def make_distinct_frames(ctx, hvc, planes, n_distinct, base_seed, rank):
    """Coefficient records of n_distinct synthetic frames via the library's own forward path
    (hvc_encode_frames on the GPU, Quant_tables.scale 75).  Returns (device int16 tensor
    [n_distinct, coef_count], qtabs uint16 [2, 64])."""
    import torch
    from video_coding_amd.synth import synth_frame_pixels
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    specs, cfs, pfs = hvc.hvc.frame_layout(planes)
    pix = np.stack([synth_frame_pixels(distinct_seed(base_seed, rank, f), planes) for f in range(n_distinct)])
    d_pix = torch.from_numpy(pix).cuda()
    d_coefs = torch.zeros((n_distinct, cfs), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.encode_frames(d_pix, pfs, qtabs, hvc.hvc.components(specs), n_distinct, d_coefs, cfs)
    ctx.synchronize()
    return d_coefs, qtabs


def _oracle_frame(orc, rec, qtabs, planes):
    off = 0
    for bw, bh, qt in planes:
        n = bw * bh * 64
        orc.dequant_idct_recon(rec[off:off + n], qtabs[qt], bw, bh)
        off += n


def cpu_baseline(frames, qtabs, planes, pixels_per_frame, min_seconds=10.0):
    """Oracle block stage (scalar C, int64, one block at a time) on the same frames: (i) one thread
    -- the model's CPU path as restated; (ii) the same code frame-sharded over the host cores this
    process may use (ctypes releases the GIL), SURVEY.md 8(d)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orc
    done, t0 = 0, time.perf_counter()
    while True:
        _oracle_frame(orc, frames[done % len(frames)], qtabs, planes)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # a one-GPU box's CPU share is 16 cores
    per_thread = max(4, int(done / dt * min_seconds / 2))  # about min_seconds/2 of wall time

    def worker(k):
        for i in range(per_thread):
            _oracle_frame(orc, frames[(k + i) % len(frames)], qtabs, planes)

    t1 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(worker, range(cores)))
    dtm = time.perf_counter() - t1
    return {"value": round(done * pixels_per_frame / dt / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d frames of the same workload, %.1f s, oracle/hvc_oracle.c "
                      "orc_dequant_idct_recon, 1 thread" % (done, dt),
            "all_cores": {"value": round(cores * per_thread * pixels_per_frame / dtm / 1e6, 3), "unit": "Mpixel/s", "cores": cores,
                          "sample": "%d frames, %.1f s, frame-sharded threads" % (cores * per_thread, dtm)}}


DOMINANT_KERNEL = "k_decode_packed"   # what this file's step launches (hvc_decode_frames, default kernel choice)


def step_kernel():
    """the kernel symbol the config 2 / 4 step really launches: HVC_DECODE_KERNEL in the environment (A/B runs only) makes
    hvc_decode_frames launch another one, and a counter pass of k_decode_packed is not that kernel's"""
    v = os.environ.get("HVC_DECODE_KERNEL", "")
    return "k_decode_q16" if v[:1] == "q" else "k_decode_fast" if v[:2] == "v2" else DOMINANT_KERNEL


def running_build():
    """the kernel id of the library this process runs (hvc_version: the hash of the kernel sources it was built from)"""
    try:
        import video_coding_amd as hvc
        return hvc.hvc.kernel_build_id()
    except Exception:
        return None


def measured_traffic(config, frames, kernel=DOMINANT_KERNEL, build="running"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate --pmc runs of this very
    command and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot collect counters
    itself: (None, reason) when the profile is absent, was taken at another configuration or launch
    size, belongs to another kernel than the one this run launched (an entry names its kernel
    symbol; one that does not is not trusted), or was taken on ANOTHER BUILD of the kernels than the
    one running now (an entry carries the kernel id of hvc_version(); one without it is stale)."""
    if build == "running":
        build = running_build()
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        stale = None
        for e in t.get("entries", [t]):
            if e.get("frames_per_launch") == frames and e.get("config", 2) == config:
                if not str(e.get("kernel", "")).startswith(kernel):
                    stale = stale or "profiles/traffic.json has a pass for this configuration, but of kernel %r, not %r: not reported" % (e.get("kernel"), kernel)
                    continue
                if e.get("build") != build:
                    stale = stale or "stale: profiled build %s (session %s), running build %s: not reported" % (e.get("build"), e.get("session"), build)
                    continue
                return round(e["hbm_bytes"]), "profiles/traffic.json (session %s, build %s, kernel %s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
                                              "%s, not collected in this run" % (e.get("session"), e["build"], e["kernel"],
                                                                                 "this command" if config in (2, 4) else e.get("source", "the same workload"))
        return None, stale or "no committed PMC pass for config %d at %d frames per launch" % (config, frames)
    except (OSError, ValueError, KeyError):
        return None, "profiles/traffic.json absent"


def other_measurements(threads):
    """`others` of the JSON line: the figures DESIGN.md quotes beside the headline, measured in this very run AFTER everything
    `value` is made of (tools/bench_configs.py's functions, called, not shelled out; each K5-verified against
    tests/golden/bench_checksums.json).  Kernels: ms per launch and the fraction of 8 TB/s their algorithmic bytes make;
    pipelines: Gpixel/s and what binds them.  Never part of `value`."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    cpus = len(os.sched_getaffinity(0))
    cpu_quota = cgroup_cpu_quota()
    cpu_quota = None if cpu_quota is None else round(cpu_quota, 2)
    out = {"host": {"cpus_in_affinity_mask": cpus, "cgroup_cpu_quota": cpu_quota, "threads_used": threads}}

    def run(name, fn, bound, pick):
        t0 = time.perf_counter()
        try:
            r = fn()
            e = pick(r)
            e["bound"] = bound
            e["verified"] = (r.get("checksum") or {}).get("verified")
        except Exception as ex:  # (one entry failing must not take the headline line with it; the entry says so)
            e = {"error": "%s: %s" % (type(ex).__name__, ex)}
        e["wall_s"] = round(time.perf_counter() - t0, 1)
        out[name] = e
        torch.cuda.empty_cache()

    def with_traffic(e, config, frames, kernel, algo_bytes):
        """the committed PMC pass of this kernel at this launch size (profiles/traffic.json), under the headline's rule: an entry
        counts only when it names the kernel symbol that ran"""
        e["traffic"], e["traffic_source"] = measured_traffic(config, frames, kernel)
        e["algorithmic_bytes_per_launch"] = algo_bytes
        if e["traffic"]:
            e["traffic_over_algorithmic"] = round(e["traffic"] / algo_bytes, 4)
        return e

    kern = lambda r: {"ms": r["kernel_ms"], "frac_of_8TBps": r["frac_of_8TBps"], "algorithmic_GBps": r["algorithmic_GBps"]}
    pipe = lambda r: {"Gpixel_s": round(r["value"] / 1e3, 2), "frames": r["frames"], "wall_ms": r["wall_ms"], "threads": r["host_threads"],
                      "frames_per_chunk": r["frames_per_chunk"]}
    run("config4_launch_128x4K444", lambda: bc.resident_decode(bc.make_args(frames=128, steps=32, warmup=16),
                                                               [(480, 270, 0), (480, 270, 1), (480, 270, 1)], 3840, 2160, 4), "hbm", kern)
    run("k3_encode_256x4K420", lambda: bc.config5(bc.make_args(frames=256, steps=40, warmup=20)), "hbm",
        lambda r: with_traffic(kern(r), 5, 256, "k_encode", 256 * 194400 * ALGO_BYTES_PER_BLOCK))
    run("fused444_512x1080p", lambda: bc.config_444(bc.make_args(frames=512, steps=40, warmup=20, fused_only=True)), "hbm",
        lambda r: with_traffic({"ms": r["fused_ms"], "frac_of_8TBps": r["frac_of_8TBps"], "algorithmic_GBps": r["algorithmic_GBps"]},
                               7, 512, "k_decode_444", r["algorithmic_bytes"]))
    run("k2_upsample420_512_planes", lambda: bc.config_k2(bc.make_args(frames=256, steps=100, warmup=50)), "hbm", kern)
    run("subsample420_512_planes", lambda: bc.config_sub420(bc.make_args(frames=256, steps=100, warmup=50)), "hbm", kern)
    # the int64 kernel as a whole call: hvc_set_decode_kernel(ctx, 2), and a DQT entry above 255
    wide = lambda r: dict(kern(r), Gpixel_s=round(r["value"] / 1e3, 2), frames=r["frames"], wide_path_blocks=r["wide_path_blocks"])
    run("wide_only_64x1080p", lambda: bc.config_wide(bc.make_args(frames=64, steps=20, warmup=5, wide_mode="kernel2")), "valu (int64)", wide)
    run("wide_dqt16_64x1080p", lambda: bc.config_wide(bc.make_args(frames=64, steps=20, warmup=5, wide_mode="dqt16")), "valu (int64)", wide)
    # ... and the exactness contract's worst case under 8-bit tables: adversarial records on which every block fails the packed
    # kernel's guard and goes through the fix-up list (whole call: packed kernel + list + k_decode_wide)
    run("fixup_worst_case_64x1080p", lambda: bc.config_fixup(bc.make_args(frames=64, steps=10, warmup=3)), "valu (int64) + one list",
        lambda r: {"ms": r["ms_per_call"], "frac_of_8TBps": r["frac_of_8TBps"], "Gpixel_s": round(r["value"] / 1e3, 2), "frames": r["frames"],
                   "wide_path_blocks": r["wide_path_blocks"]})
    # the asynchronous seam: the CALLER's reader fills pinned slots, hvc_decode_frames_submit / hvc_wait
    seam = lambda r: dict(pipe(r), h2d_GBps=r["h2d_GBps"], d2h_GBps=r["d2h_GBps"], refill_GBps=r["refill_GBps"],
                          overlap_fraction=r["overlap_fraction"], slots=r["slots"], frames_per_slot=r["frames_per_slot"],
                          records_verified=r["checksum"]["records"])
    run("async_seam_4096x1080p_to_hbm", lambda: bc.config_async(bc.make_args(frames=4096, steps=2, threads=threads, chunk=64)), "pcie", seam)
    run("async_seam_2048x1080p_to_pinned_host", lambda: bc.config_async(bc.make_args(frames=2048, steps=2, threads=threads, chunk=64, host_out=True)),
        "pcie", seam)
    # BASELINE config 3 at its own size (4096 x 1080p files), config 5 end to end on 256 x 4K frames
    pipe3 = lambda r: dict(pipe(r), h2d_GBps=r["h2d_GBps"], overlap_fraction=r["overlap_fraction"])
    run("config3_host_reader_4096_files", lambda: bc.config3(bc.make_args(frames=4096, steps=2, threads=threads, chunk=32)), "host", pipe3)
    run("config3_gpu_reader_4096_files", lambda: bc.config3(bc.make_args(frames=4096, steps=2, threads=threads, chunk=256, gpu_entropy=True)),
        "pcie", pipe3)
    run("config5_files_host_coder_256_frames", lambda: bc.config5_files(bc.make_args(frames=256, steps=2, threads=threads, chunk=16)), "host", pipe)
    run("config5_files_gpu_coder_256_frames", lambda: bc.config5_files(bc.make_args(frames=256, steps=2, threads=threads, chunk=16, gpu_entropy=True)),
        "pcie", pipe)
    return out


def expected_checksums(config, n_distinct, rank):
    """tests/golden/bench_checksums.json: the K5 checksums of the decoded distinct frames as the model
    restatement gives them (made by tests/golden/make_bench_checksums.py, re-derived by the CPU suite)."""
    try:
        # (HVC_BENCH_GOLDEN: another file -- how the suite checks that a mismatch fails the command)
        with open(os.environ.get("HVC_BENCH_GOLDEN") or os.path.join(ROOT, "tests", "golden", "bench_checksums.json")) as f:
            g = json.load(f)
        vals = g["bench_config%d" % config]["rank%d" % rank]
        return [int(v, 16) for v in vals[:n_distinct]] if len(vals) >= n_distinct else None
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
    if use_group(world) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver (a rank started by
        # another launcher than this file's own: the variable is normally exported already; never overridden)
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return dist


def use_group(world):
    """a process group exists: more than one rank -- or one rank started by a launcher with HVC_BENCH_DIST_ALWAYS=1,
    which lets a one-GPU box run the RCCL calls of the N > 1 path (init, barrier, all-reduce) for real"""
    return world > 1 or (os.environ.get("HVC_BENCH_DIST_ALWAYS") == "1" and "WORLD_SIZE" in os.environ)


def timed_steps(step, steps, warmup, sync, world, dist=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + sync on both sides."""
    for _ in range(warmup):
        step()
    sync()
    if use_group(world):
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if use_group(world):
        dist.barrier()
    sync()
    return time.perf_counter() - t0


def sustained_run(step, seconds, launches_per_step):
    """>= `seconds` of the same step back to back AFTER the timed region (so `value` / `ms_per_step` are untouched):
    every step bracketed by its own pair of events on the stream the library launches on (the context was put on
    torch's current stream), enqueued in batches so the host stays ahead of the GPU without an unbounded queue.
    Returns the `sustained` object of the JSON line: a drifting clock shows as first != last decile."""
    import torch
    if seconds <= 0:
        return None
    events, t0 = [], time.perf_counter()
    batch = max(1, 64 // launches_per_step)
    while time.perf_counter() - t0 < seconds:
        for _ in range(batch):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            events.append((e0, e1))
        events[-1][1].synchronize()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = np.array([a.elapsed_time(b) for a, b in events])
    dec = max(1, len(ms) // 10)
    return {"steps": len(ms), "launches": len(ms) * launches_per_step, "wall_s": round(wall, 3),
            "first_decile_ms": round(float(ms[:dec].mean()), 4), "last_decile_ms": round(float(ms[-dec:].mean()), 4),
            "median_ms": round(float(np.median(ms)), 4), "min_ms": round(float(ms.min()), 4), "max_ms": round(float(ms.max()), 4),
            "gpu_busy_fraction": round(float(ms.sum()) * 1e-3 / wall, 3),
            "what": "the same step repeated after the timed region, one event pair per step; not part of value"}


def max_over_ranks(dt, world, dist=None, device="cpu"):
    if not use_group(world):
        return dt
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(values, world, dist=None, device="cpu"):
    """every rank's list of floats on every rank ([rank][k]): the per-rank kernel times of rank 0's line, so that a slow
    GPU shows in the scaling record instead of hiding inside the MAX over ranks.  Timing closure like the barrier: the
    data path has no collective."""
    if not use_group(world):
        return [list(values)]
    import torch
    mine = torch.tensor(list(values), dtype=torch.float64, device=device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return [[float(x) for x in p.tolist()] for p in parts]


def parse_cpulist(text):
    """'0-15,32-47' -> [0, ..., 15, 32, ..., 47] (the format of sysfs local_cpulist and of hvc_get_host_cpus)"""
    cpus = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def bind_rank_to_gpu_node(ctx):
    """N > 1 on one node: the rank's launch thread and the context's host threads on the CPUs of the GPU's NUMA node
    (hvc_set_host_cpus "auto" = the local_cpulist of the GPU's PCI function, inside the process's own mask) -- a step of
    config 4 is 16 launches of 1.6 ms, little slack for a launch thread on the other socket.  Returns the CPU list as text,
    or None where the node cannot be told (then nothing is changed)."""
    try:
        ctx.set_host_cpus("auto")
        text, n = ctx.get_host_cpus()
        cpus = parse_cpulist(text) if n > 0 else []
        if cpus:
            os.sched_setaffinity(0, cpus)
            return text
    except Exception:
        pass
    return None


def format_cpulist(cpus):
    """[0, 1, 2, 8, 9] -> '0-2,8-9' (what hvc_set_host_cpus takes)"""
    cpus, parts, i = sorted(cpus), [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append("%d" % cpus[i] if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(parts)


def host_share(node_keys, rank, cpus, cap=16):
    """config 3 at N > 1: the host Huffman threads of the ranks whose GPUs hang off one NUMA node share that node's CPUs.
    node_keys[r] names rank r's node (its first CPU); `cpus` = this rank's node CPUs.  -> (this rank's CPUs, its thread count):
    an equal contiguous slice per rank on the node, `cap` threads at most (a one-GPU box's CPU share), one at least."""
    peers = [r for r, k in enumerate(node_keys) if k == node_keys[rank]]
    cpus = sorted(cpus)
    share = max(1, len(cpus) // len(peers))
    i = peers.index(rank)
    mine = cpus[i * share:(i + 1) * share] or cpus[-1:]
    return mine, max(1, min(cap, len(mine)))


def cgroup_cpu_quota():
    """the container's CPU quota (cgroup v2 cpu.max: "quota period") as a number of CPUs, None where there is none: what the host
    stages really have, whatever the affinity mask shows (a one-GPU box of the pool: a mask of 256 CPUs, a quota of 16)"""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else int(quota) / int(period)
    except (OSError, ValueError):
        return None


def share_host_cpus(ctx, rank, world, dist, device, bind=True):
    """-> (cpulist text or None, host threads): at N > 1 the rank's launch thread and its pipeline threads are put on the rank's
    slice of its GPU's NUMA node (host_share); at N = 1 nothing is bound and the threads are min(16, CPUs of the mask)."""
    mask = sorted(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()   # (the ranks of one node share one container: its quota is split between them)
    cap = 16 if quota is None else max(1, min(16, int(quota / world)))
    if world == 1:
        return None, max(1, min(cap, len(mask)))
    cpus = mask
    if ctx is not None and bind:
        try:
            ctx.set_host_cpus("auto")
            text, n = ctx.get_host_cpus()
            if n > 0:
                cpus = parse_cpulist(text)
        except Exception:
            pass
    keys = [int(k[0]) for k in gather_over_ranks([float(cpus[0])], world, dist, device)]
    mine, threads = host_share(keys, rank, cpus, cap)
    text = format_cpulist(mine)
    if ctx is not None and bind:
        try:
            ctx.set_host_cpus(text)
            os.sched_setaffinity(0, mine)
        except Exception:
            text = None
    return text, threads


def whole_job_mpixels(world, frames_per_gpu, steps, dt, pixels_per_frame=W * H):
    """value = units ALL ranks processed / max-over-ranks time (weak scaling: per-GPU work fixed)."""
    return world * frames_per_gpu * steps * pixels_per_frame / dt / 1e6


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` typed by hand: N fresh ranks through torch.distributed.run, started as
    CHILD processes before this one has touched HIP or torch.cuda (a process that has must never be
    replaced or re-executed); the ranks inherit stdout, so rank 0's JSON line is this command's output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)  # the first launches after idle run off-clock (DVFS)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per launch (9.6 GB of coefficients + pixels)")
    ap.add_argument("--shard", type=int, default=None, help="config 4: frames per GPU per step (launches of --frames)")
    ap.add_argument("--distinct", type=int, default=None, help="distinct synthetic frames (replicated): 8, configs 3 / 5: 4")
    ap.add_argument("--threads", type=int, default=None, help="config 3: host threads per rank (default: the rank's share of its NUMA node, <= 16)")
    ap.add_argument("--tight", action="store_true", help="planes and frames back to back instead of on 64 KiB / 2 MiB boundaries (A/B)")
    ap.add_argument("--no-other-layout", action="store_true", help="skip roofline.other_layout (the same frames in the other layout, 40 launches after the timed region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-others", action="store_true", help="skip `others` (configs 3 / 4 / 5 and the other kernels, after the timed region)")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="after the K timed steps: this many seconds of the same step back to back, reported as "
                         "`sustained` (never part of `value`); 0 = skip")
    args = ap.parse_args(argv)
    wl = WORKLOADS[args.config]
    args.frames = args.frames or wl["frames"]
    args.distinct = args.distinct or (4 if args.config in (3, 5) else 8)
    if args.config == 3:
        if args.frames % args.gpus:
            ap.error("--config 3: --frames (files in total) must be a multiple of --gpus")
        args.frames //= args.gpus   # files per rank: the batch is split, strong scaling
    args.shard = args.shard or wl["shard"] or args.frames
    if args.shard % args.frames:
        ap.error("--shard must be a multiple of --frames")
    args.steps = args.steps if args.steps is not None else wl["steps"]
    args.warmup = args.warmup if args.warmup is not None else (10 if args.config in (2, 5) else 1)
    return args


def run_without_gpu(args, rank, world):
    """HVC_BENCH_NO_GPU=1: the launch path alone (self-launch, rendezvous, barriers, MAX over ranks, one JSON
    line from rank 0) with a sleep standing in for the step -- what tests/test_distributed_cpu.py drives on a
    machine without a GPU.  Nothing is decoded and the line says so; it is never a measurement."""
    dist = dist_init(world, "gloo")
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.001 * (rank + 1))

    dt_local = timed_steps(step, args.steps, args.warmup, lambda: None, world, dist)
    dt = max_over_ranks(dt_local, world, dist, "cpu")
    step_ms = 1.0 * (rank + 1)   # (what stands in for a rank's kernel time here: its sleep)
    per_rank = gather_over_ranks([step_ms, step_ms, step_ms], world, dist, "cpu")
    # config 3: the ranks' host-thread shares, computed as on a GPU box (all ranks on this machine's one "node": its CPU mask)
    cpus_text, threads = share_host_cpus(None, rank, world, dist, "cpu") if args.config == 3 else (None, None)
    shares = gather_over_ranks([float(threads)], world, dist, "cpu") if args.config == 3 else None
    if rank == 0:
        wl = WORKLOADS[args.config]
        line = {"metric": wl["metric"], "value": round(whole_job_mpixels(world, args.shard, args.steps, dt, wl["W"] * wl["H"]), 1),
                "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if args.config == 3 else "weak",
                "vs_baseline": None, "dtype": "int32",
                "data": "none: launch-path rehearsal without a GPU (HVC_BENCH_NO_GPU=1), nothing decoded, not a measurement",
                "config": {"workload": wl["name"], "baseline_config": args.config, "step_calls_rank0": len(calls)},
                "per_rank_kernel_ms": {"mean_min_max": [[round(x, 4) for x in r] for r in per_rank],
                                       "what": "launch-path rehearsal: each rank's sleep per step"}}
        if args.config == 3:
            line["config"].update({"files_total": args.frames * world, "files_per_gpu_per_step": args.frames,
                                   "host_threads_per_rank": [int(x[0]) for x in shares], "rank0_cpus": cpus_text})
        print(json.dumps(line), flush=True)
    if use_group(world):
        dist.destroy_process_group()


def all_ranks_ok(ok_local, world, dist, device):
    """how many ranks verified their own output (every rank knows the answer: it comes out of an all-reduce)"""
    if not use_group(world):
        return int(bool(ok_local))
    import torch
    t = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def cpu_baseline_files(jpegs, pixels_per_frame, min_seconds=10.0):
    """config 3's CPU baseline: the model's whole CPU path on the same files -- Decoder.decode_a_frame as restated in
    oracle/hvc_oracle.c (bit-at-a-time Huffman reader, block stage, crop), scalar, one thread, a bounded sample."""
    from oracle import orc
    done, t0 = 0, time.perf_counter()
    while True:
        orc.decode_a_frame(jpegs[done % len(jpegs)])
        done += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    return {"value": round(done * pixels_per_frame / dt / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d of the same 1080p files, %.1f s, oracle/hvc_oracle.c decode_a_frame (Huffman + block stage), 1 thread" % (done, dt)}


def run_files(args, rank, world, local_rank, rehearsal):
    """--config 3: BASELINE.json configs[2] over N ranks (decoder.ml:118-140, 261-281 stay on the host; :142-149, 213-224 and
    dct.ml:11-107 on the GPU)."""
    import torch
    import video_coding_amd as hvc
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    wl = WORKLOADS[3]
    PW, PH = wl["W"], wl["H"]
    torch.cuda.set_device(local_rank)
    dev = "cpu" if rehearsal else "cuda"
    dist = dist_init(world, "gloo" if rehearsal else "nccl", torch.device("cuda", local_rank))
    ctx = hvc.Context(local_rank)
    cpus_text, threads = share_host_cpus(ctx, rank, world, dist, dev, bind=not rehearsal or world > 1)
    if args.threads:
        threads = args.threads
    jpegs = bc.config3_files(ctx, args.distinct)   # the library's own encoder writes the input files, outside any timed region
    n = args.frames
    batch = [jpegs[i % len(jpegs)] for i in range(n)]
    info = hvc.hvc.jpeg_read_header(batch[0])
    d_pix = torch.zeros(n * info.pixel_bytes, dtype=torch.uint8, device="cuda")
    blocks_per_frame = sum(bw * bh for bw, bh, _ in wl["planes"])
    readers = {}
    for reader in ("host", "gpu"):
        gpu = reader == "gpu"
        chunk = 0 if gpu else 32   # (0: the library's own choice for the GPU reader -- a quarter of the batch, 64..256)
        d_pix.zero_()
        ctx.jpeg_decode_batch(batch[:min(64, n)], d_pix, info.pixel_bytes, threads=threads, frames_per_chunk=chunk, gpu_entropy=gpu)
        d_pix.zero_()
        torch.cuda.synchronize()
        stats = []

        def step():
            stats.append(ctx.jpeg_decode_batch(batch, d_pix, info.pixel_bytes, threads=threads, frames_per_chunk=chunk, gpu_entropy=gpu))

        dt_local = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world, dist)
        dt = max_over_ranks(dt_local, world, dist, dev)
        timed = stats[-args.steps:]
        chk = bc.verify(ctx, d_pix, info.pixel_bytes, n, "configs_c3", args.distinct)["checksum"]
        ranks_ok = all_ranks_ok(chk["verified"] is True, world, dist, dev)
        k_ms = sum(st.kernel_ms_sum for st in timed) / len(timed)
        e_ms = sum(st.entropy_ms_sum for st in timed) / len(timed)
        h_ms = sum(st.h2d_ms_sum for st in timed) / len(timed)
        per_rank = gather_over_ranks([dt_local / args.steps * 1e3, e_ms, h_ms, k_ms, float(threads)], world, dist, dev)
        readers[reader] = {
            "value": round(world * n * args.steps * PW * PH / dt / 1e6, 1), "unit": "Mpixel/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
            "timed_region_s": round(dt, 4), "frames_per_chunk": timed[-1].frames_per_chunk, "chunks": timed[-1].chunks,
            "h2d_MB_per_step": round(timed[-1].coef_bytes / 1e6, 1),
            "h2d_GBps": round(timed[-1].coef_bytes / (max(h_ms, 1e-9) * 1e-3) / 1e9, 1),
            # how much of the stages' time the pipeline hides: 1 - wall / (entropy / threads + upload + kernels)
            "overlap_fraction": round(1.0 - (dt_local / args.steps * 1e3) / max(e_ms / threads + h_ms + k_ms, 1e-9), 3),
            "kernel_launch_frames": timed[-1].frames_per_chunk,
            "verified": chk["verified"], "ranks_verified": ranks_ok,
            "checksum_rank0": chk["distinct"],
            "per_rank": {"wall_ms_entropy_thread_ms_sum_h2d_ms_kernel_ms_threads": [[round(x, 2) for x in r] for r in per_rank],
                         "slowest_over_fastest": round(max(r[0] for r in per_rank) / min(r[0] for r in per_rank), 4)},
            "gpu_busy_fraction": round(k_ms / (dt / args.steps * 1e3), 4),
            "bound": ("pcie: unstuffed segments up + GPU reader kernels" if gpu else
                      "host: Huffman thread time / threads ~ wall (%.0f ms / %d = %.0f ms)" % (e_ms, threads, e_ms / threads)),
            "k_decode_packed_algorithmic_GBps": round(n * blocks_per_frame * ALGO_BYTES_PER_BLOCK / (max(k_ms, 1e-9) * 1e-3) / 1e9, 1)}
    ok = all(r["ranks_verified"] == world for r in readers.values())
    if rank == 0:
        host, gpu_r = readers["host"], readers["gpu"]
        # the chunk-sized k_decode_packed launches of the host-reader pipeline: a committed PMC pass of this command
        traffic, traffic_source = measured_traffic(3, host["kernel_launch_frames"], "k_decode_packed")
        chunk_algo = host["kernel_launch_frames"] * blocks_per_frame * ALGO_BYTES_PER_BLOCK
        out = {"metric": wl["metric"], "value": host["value"], "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": host["ms_per_step"], "timed_region_s": host["timed_region_s"],
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int32",
               "data": "synthetic: %d distinct seeded 1080p frames written as baseline JPEG files by hvc_jpeg_encode (q75), replicated"
                       % args.distinct + (" (REHEARSAL: all ranks share cuda:0, gloo)" if rehearsal else ""),
               "config": {"workload": "%s; %d files in total, %d per GPU per step" % (wl["name"], n * world, n), "baseline_config": 3,
                          "files_total": n * world, "files_per_gpu_per_step": n, "host_threads_per_rank": threads,
                          "cgroup_cpu_quota": cgroup_cpu_quota(),
                          "rank0_cpus": cpus_text, "blocks_per_frame": blocks_per_frame,
                          "parallelism": "the file batch split over the GPUs, no collective; host threads = the rank's share of its GPU's NUMA node"},
               "roofline": {"bound": "hbm", "kernel": "k_decode_packed", "achieved": host["k_decode_packed_algorithmic_GBps"],
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(host["k_decode_packed_algorithmic_GBps"] / HBM_PEAK_GBPS, 4),
                            "traffic": traffic, "traffic_source": traffic_source,
                            "algorithmic_bytes_per_launch": chunk_algo,
                            "traffic_over_algorithmic": round(traffic / chunk_algo, 4) if traffic else None,
                            "h2d_GBps": host["h2d_GBps"], "overlap_fraction": host["overlap_fraction"],
                            "note": "chunk-sized launches inside a HOST-bound pipeline (GPU busy %.1f %% of the step): the kernel's own "
                                    "roofline figure is config 2's line" % (100 * host["gpu_busy_fraction"])},
               "checksum": {"kernel": "k_checksum (K5, hvc_checksum_records) over EVERY decoded frame", "verified": host["verified"],
                            "ranks_verified": host["ranks_verified"], "expected": "tests/golden/bench_checksums.json:configs_c3"},
               "host_reader": host, "gpu_reader": gpu_r}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_files(jpegs, PW * PH, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_group(world):
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


def run_encode(args, rank, world, local_rank, rehearsal):
    """--config 5: BASELINE.json configs[4], the encoder's block stage (encoder.ml:81-108, dct.ml:109-196) on HBM-resident frames"""
    import torch
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_frame_pixels
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as bc
    wl = WORKLOADS[5]
    planes, PW, PH = wl["planes"], wl["W"], wl["H"]
    torch.cuda.set_device(local_rank)
    dev = "cpu" if rehearsal else "cuda"
    dist = dist_init(world, "gloo" if rehearsal else "nccl", torch.device("cuda", local_rank))
    ctx = hvc.Context(local_rank)
    node_cpus = bind_rank_to_gpu_node(ctx) if (world > 1 and not rehearsal) else None
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    qtabs = np.stack([hvc.hvc.quant_table(0, 75), hvc.hvc.quant_table(1, 75)])
    tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
    align = 1 if args.tight else hvc.hvc.layout_alignment(planes)   # (as in config 2: planes and frames on 2 MiB boundaries here)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes, align=align)
    comps = hvc.hvc.components(specs)
    recs = np.stack([synth_frame_pixels(60 + 8 * f, planes) for f in range(args.distinct)])   # (tools/bench_configs.py config5's seeds)
    reps = (args.frames + args.distinct - 1) // args.distinct
    d_pix = hvc.hvc.spread_records(torch.from_numpy(recs).cuda(), tspecs, specs, pfs, "plane_offset").repeat(reps, 1)[:args.frames].contiguous()
    d_coefs = torch.zeros((args.frames, cfs), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()

    def step():
        ctx.encode_frames(d_pix, pfs, qtabs, comps, args.frames, d_coefs, cfs)

    for _ in range(8):   # setup, not measurement: pages touched, clock ramped
        step()
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    dt = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world, dist)
    dt = max_over_ranks(dt, world, dist, dev)
    kernel_ms = ctx.kernel_ms_history(min(args.steps, 64))
    chk = bc.verify(ctx, hvc.hvc.tight_records(d_coefs, specs, "coef_offset"), tcfs * 2, args.frames, "configs_c5", args.distinct)["checksum"]
    ranks_ok = all_ranks_ok(chk["verified"] is True, world, dist, dev)
    per_rank = gather_over_ranks([float(np.mean(kernel_ms)), float(np.min(kernel_ms)), float(np.max(kernel_ms))], world, dist, dev)
    if rank == 0:
        blocks_per_frame = sum(bw * bh for bw, bh, _ in planes)
        k_ms = float(np.mean(kernel_ms))
        algo_bytes = args.frames * blocks_per_frame * ALGO_BYTES_PER_BLOCK
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(5, args.frames, "k_encode")
        out = {"metric": wl["metric"], "value": round(whole_job_mpixels(world, args.frames, args.steps, dt, PW * PH), 1), "unit": "Mpixel/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "timed_region_s": round(dt, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
               "data": "synthetic" + (" (REHEARSAL: all ranks share cuda:0, gloo)" if rehearsal else ""),
               "config": {"workload": "%s, %d frames/GPU/step" % (wl["name"], args.frames), "baseline_config": 5,
                          "frames_per_gpu_per_step": args.frames, "blocks_per_frame": blocks_per_frame,
                          "layout": "planes back to back (tight)" if align == 1 else
                                    "every plane of every frame (pixels and coefficients) on a %d KiB boundary" % (align >> 10),
                          "parallelism": "independent frame batch per GPU, no collective"},
               "roofline": {"bound": "hbm", "kernel": "k_encode", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                            "kernel_ms": round(k_ms, 4), "launches_averaged": len(kernel_ms), "algorithmic_bytes_per_launch": algo_bytes},
               "checksum": {"kernel": "k_checksum (K5) over EVERY coefficient record", "verified": chk["verified"], "ranks_verified": ranks_ok,
                            "expected": "tests/golden/bench_checksums.json:configs_c5", "rank0": chk["distinct"]},
               "per_rank_kernel_ms": {"mean_min_max": [[round(x, 4) for x in r] for r in per_rank], "rank0_cpus": node_cpus}}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import orc   # (the checker, timed as the baseline only)
            frames, done, t0 = recs, 0, time.perf_counter()
            while True:
                off = 0
                for bw, bh, qt in planes:
                    nb = bw * bh * 64
                    orc.fdct_quant(frames[done % len(frames)][off:off + nb].reshape(bh * 8, bw * 8), qtabs[qt], bw, bh)
                    off += nb
                done += 1
                dtc = time.perf_counter() - t0
                if dtc >= args.cpu_seconds:
                    break
            out["cpu_baseline"] = {"value": round(done * PW * PH / dtc / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
                                   "sample": "%d frames of the same workload, %.1f s, oracle/hvc_oracle.c orc_fdct_quant, 1 thread" % (done, dtc)}
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_group(world):
        dist.destroy_process_group()
    if ranks_ok != world:
        sys.exit(3)


def main():
    args = parse_args()
    rank, world, local_rank = dist_env()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))  # (no HIP / torch.cuda call has happened in this process)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (set by the launcher): they must agree" % (args.gpus, world))
    if os.environ.get("HVC_BENCH_NO_GPU") == "1":
        return run_without_gpu(args, rank, world)
    if args.config in (3, 5):
        # (rehearsal on a one-GPU box: every rank on cuda:0, gloo -- never a measurement, the line says so)
        rehearsal = os.environ.get("HVC_BENCH_REHEARSAL") == "1"
        return (run_files if args.config == 3 else run_encode)(args, rank, world, 0 if rehearsal else local_rank, rehearsal)

    import torch
    import video_coding_amd as hvc

    wl = WORKLOADS[args.config]
    planes, PW, PH = wl["planes"], wl["W"], wl["H"]
    blocks_per_frame = sum(bw * bh for bw, bh, _ in planes)
    # Rehearsal on a one-GPU box (never the measured configuration): HVC_BENCH_REHEARSAL=1 puts every
    # rank on cuda:0 and closes the timing with gloo instead of RCCL (which refuses two ranks per GPU).
    rehearsal = os.environ.get("HVC_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = dist_init(world, "gloo" if rehearsal else "nccl", torch.device("cuda", local_rank))

    ctx = hvc.Context(local_rank)  # raises without a gfx950 GPU: there is no CPU fallback
    node_cpus = bind_rank_to_gpu_node(ctx) if (world > 1 and not rehearsal) else None
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    d_distinct, qtabs = make_distinct_frames(ctx, hvc, planes, args.distinct, wl["seed"], rank)   # (tight records)
    # The resident batch: every plane of every frame, coefficients and pixels, on a 64 KiB (1080p) / 2 MiB (4K) boundary -- the
    # caller's choice of hvc_component offsets and frame strides; +0.4 ... +1.0 points over planes back to back
    # (profiles/r05l_alignment_sweep.txt).  Padding is neither read nor written; K5 runs on the planes gathered tight.
    tspecs, tcfs, tpfs = hvc.hvc.frame_layout(planes)
    align = 1 if args.tight else hvc.hvc.layout_alignment(planes)
    specs, cfs, pfs = hvc.hvc.frame_layout(planes, align=align)
    comps = hvc.hvc.components(specs)
    d_distinct_laid = hvc.hvc.spread_records(d_distinct, tspecs, specs, cfs, "coef_offset")
    launches = args.shard // args.frames
    # the shard resident as a whole when the device has the room (config 4: 153 GB per GPU); otherwise one
    # launch-sized chunk, processed `launches` times per step
    need = args.shard * (cfs * 2 + pfs)
    free_b, _ = torch.cuda.mem_get_info()
    resident = launches == 1 or (not rehearsal and need + (8 << 30) < free_b)
    held = args.shard if resident else args.frames
    d_coefs = torch.empty((held, cfs), dtype=torch.int16, device="cuda")
    for f0 in range(0, held, args.distinct):  # replicate the distinct frames
        n = min(args.distinct, held - f0)
        d_coefs[f0:f0 + n] = d_distinct_laid[:n]
    d_pix = torch.zeros((held, pfs), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.set_profiling(True)

    def step():
        for k in range(launches):
            f0 = k * args.frames if resident else 0
            ctx.decode_frames(d_coefs[f0:f0 + args.frames], cfs, qtabs, comps, args.frames, d_pix[f0:f0 + args.frames], pfs)

    # Setup, not measurement: a few launches so that the output buffer's pages are touched and the chip
    # is on its sustained clock even when the caller asks for very few warm-up steps (the first launches
    # after idle run off the sustained rate in either direction).
    for _ in range(8 if launches == 1 else 1):
        step()
    torch.cuda.synchronize()
    ctx.set_profiling(True)  # restart the event ring: only warm-up and timed steps are recorded from here

    dt = timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize, world, dist)
    dt = max_over_ranks(dt, world, dist, "cpu" if rehearsal else "cuda")

    # HIP events recorded around k_decode_packed inside the timed region (one pair per launch, ring of 64)
    kernel_ms = ctx.kernel_ms_history(min(args.steps * launches, 64))
    wide = ctx.last_wide_blocks()
    # K5: what was decoded -- the distinct frames' pixel records, checksummed where they are
    # ... EVERY record of the resident batch (record r holds distinct frame r % distinct), gathered tight a piece at a time
    def output_checksums():
        sums = []
        for f0 in range(0, held, 128):
            k = min(128, held - f0)
            sums += [int(x) for x in ctx.checksum_records(hvc.hvc.tight_records(d_pix[f0:f0 + k], specs, "plane_offset"), tpfs, k)]
        return sums

    sums = output_checksums()
    want_distinct = expected_checksums(args.config, min(args.distinct, held), rank) if args.distinct <= 8 else None
    want = [want_distinct[r % args.distinct] for r in range(held)] if want_distinct is not None else None
    sustained = sustained_run(step, args.sustain_seconds, launches)  # (after everything `value` is made of)
    if sustained is not None:  # ... and what the sustained run left behind is still the model's output
        again = output_checksums()
        sustained["output_unchanged"] = again == sums
    # The other layout beside the headline's: the same frames with planes and frames back to back -- what every
    # library entry point that lays records out itself produces -- timed by the same events right after; never part of `value`.
    other_layout = None
    if launches == 1 and world == 1 and not args.no_other_layout:
        try:
            o_align = hvc.hvc.layout_alignment(planes) if args.tight else 1
            ospecs, ocfs, opfs = hvc.hvc.frame_layout(planes, align=o_align)
            o_coefs = hvc.hvc.spread_records(d_distinct, tspecs, ospecs, ocfs, "coef_offset").repeat(
                (held + args.distinct - 1) // args.distinct, 1)[:held].contiguous()
            o_pix = torch.zeros((held, opfs), dtype=torch.uint8, device="cuda")
            ocomps = hvc.hvc.components(ospecs)
            for _ in range(10):
                ctx.decode_frames(o_coefs, ocfs, qtabs, ocomps, held, o_pix, opfs)
            torch.cuda.synchronize()
            ctx.set_profiling(True)
            for _ in range(30):
                ctx.decode_frames(o_coefs, ocfs, qtabs, ocomps, held, o_pix, opfs)
            o_ms = float(np.mean(ctx.kernel_ms_history(30)))
            o_sums = []
            for f0 in range(0, held, 128):
                k = min(128, held - f0)
                o_sums += [int(x) for x in ctx.checksum_records(hvc.hvc.tight_records(o_pix[f0:f0 + k], ospecs, "plane_offset"), tpfs, k)]
            other_layout = {"layout": "planes back to back (tight)" if o_align == 1 else "planes on %d KiB boundaries" % (o_align >> 10),
                            "kernel_ms": round(o_ms, 4), "launches_averaged": 30,
                            "frac": round(args.frames * blocks_per_frame * ALGO_BYTES_PER_BLOCK / (o_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "verified": (o_sums == want) if want is not None else None}
            del o_coefs, o_pix
        except Exception as ex:   # (an A/B beside the headline must not take the line with it)
            other_layout = {"error": "%s: %s" % (type(ex).__name__, ex)}
    per_rank = gather_over_ranks([float(np.mean(kernel_ms)), float(np.min(kernel_ms)), float(np.max(kernel_ms))], world, dist,
                                 "cpu" if rehearsal else "cuda")
    frames_host = d_distinct.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    ok_local = want is not None and sums == want
    if use_group(world):  # every rank's output is verified; rank 0 reports how many were
        t = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        ranks_ok = int(t.item())
    else:
        ranks_ok = int(ok_local)

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        algo_bytes = args.frames * blocks_per_frame * ALGO_BYTES_PER_BLOCK
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        traffic, traffic_source = measured_traffic(args.config, args.frames, step_kernel())
        out = {
            "metric": wl["metric"],
            "value": round(whole_job_mpixels(world, args.shard, args.steps, dt, PW * PH), 1),
            "unit": "Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "timed_region_s": round(dt, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic" + (" (REHEARSAL: all ranks share cuda:0, gloo)" if rehearsal else ""),
            "config": {"workload": "%s, %d frames/GPU/step%s" % (
                           wl["name"], args.shard,
                           "" if launches == 1 else " in %d launches of %d frames, %s" % (
                               launches, args.frames, "whole shard resident in HBM (%.0f GB/GPU)" % (need / 1e9) if resident
                               else "one resident %d-frame chunk re-used (the shard's %.0f GB did not fit next to other users of the device)"
                               % (args.frames, need / 1e9))),
                       "baseline_config": args.config,
                       "frames_per_gpu_per_step": args.shard, "frames_per_launch": args.frames,
                       "blocks_per_frame": blocks_per_frame,
                       "parallelism": "independent frame batch per GPU, no collective",
                       "layout": "planes back to back (tight)" if align == 1 else
                                 "every plane of every frame (coefficients and pixels) on a %d KiB boundary" % (align >> 10),
                       "wide_path_blocks": int(wide)},
            "roofline": {"bound": "hbm", "kernel": step_kernel(), "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(k_ms, 4),
                         "kernel_ms_min_max": [round(float(np.min(kernel_ms)), 4), round(float(np.max(kernel_ms)), 4)],
                         "launches_averaged": len(kernel_ms),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "arithmetic": "int32 with int16-pair dot products (v_dot2_i32_i16); int64 fix-up kernel "
                                       "for blocks outside the proven range",
                         "other_layout": other_layout},
            "checksum": {"kernel": "k_checksum (K5, hvc_checksum_records) over EVERY record of the resident batch", "frames": len(sums),
                         "distinct": min(args.distinct, held), "rank0": ["%016x" % s for s in sums[:args.distinct]],
                         "expected": "tests/golden/bench_checksums.json" if want is not None else None,
                         "verified": bool(ok_local) if want is not None else None,
                         "ranks_verified": ranks_ok if want is not None else None,
                         "verification": "compared on every rank" if want is not None else
                                         "SKIPPED: no golden checksums for this run (--distinct > 8 or no entry for this rank)"},
        }
        out["per_rank_kernel_ms"] = {"mean_min_max": [[round(x, 4) for x in r] for r in per_rank],
                                     "slowest_over_fastest": round(max(r[0] for r in per_rank) / min(r[0] for r in per_rank), 4),
                                     "rank0_cpus": node_cpus,
                                     "what": "k_decode_packed's event-timed launches inside the timed region, every rank's own"}
        if sustained is not None:
            out["sustained"] = sustained
    ctx.close()
    if rank == 0:
        if world == 1 and args.config == 2 and not args.no_others and not rehearsal:
            del d_coefs, d_pix, d_distinct   # (the headline's 9.6 GB: the other workloads bring their own)
            torch.cuda.empty_cache()
            out["others"] = other_measurements(min(16, len(os.sched_getaffinity(0))))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(frames_host, qtabs, planes, PW * PH, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if use_group(world):
        dist.destroy_process_group()
    # a run whose output is not the model's is not a measurement: the line above says so ("verified": false) and the
    # command fails (every rank knows ranks_ok: it came out of an all-reduce)
    if want is not None and ranks_ok != world:
        sys.exit(3)


if __name__ == "__main__":
    main()
