"""The N>1 control flow of bench.py (one process per GPU, barrier on both sides of
the timed region, MAX over ranks, rank 0 reports the whole-job aggregate)
rehearsed on CPU with gloo, world_size 2.  The data path has no collective, so
this is all the distributed logic there is."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    rank, world, local_rank = bench.dist_env()
    dist = bench.dist_init(world, "gloo")
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.002 * (rank + 1))      # rank 1 is the slow one
    dt_local = bench.timed_steps(step, steps=5, warmup=2, sync=lambda: None, world=world, dist=dist)
    dt = bench.max_over_ranks(dt_local, world, dist, "cpu")
    with open(os.path.join(os.environ["HVC_TEST_OUT"], "rank%%d.json" %% rank), "w") as f:
        json.dump({"rank": rank, "world": world, "calls": len(calls), "dt_local": dt_local, "dt": dt,
                   "value": bench.whole_job_mpixels(world, 4, 5, dt)}, f)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_timing_closure(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HVC_TEST_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in (0, 1)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert all(r["calls"] == 7 for r in recs)                      # 2 warm-up + exactly 5 timed
    assert recs[0]["dt"] == recs[1]["dt"]                          # MAX over ranks, same on both
    assert recs[0]["dt"] >= max(r["dt_local"] for r in recs) - 1e-9
    assert recs[1]["dt_local"] >= 5 * 0.004                        # the slow rank bounds the job
    # the barrier makes the fast rank wait: its local bracket also covers the slow rank's steps
    assert recs[0]["dt_local"] >= 5 * 0.004 * 0.9
    want = 2 * 4 * 5 * 1920 * 1080 / recs[0]["dt"] / 1e6            # whole job: both ranks' frames / max time
    assert abs(recs[0]["value"] - want) < 1e-6 * want


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    n = []
    dt = bench.timed_steps(lambda: n.append(1), steps=3, warmup=1, sync=lambda: None, world=1)
    assert len(n) == 4 and dt >= 0
    assert bench.max_over_ranks(dt, 1) == dt
    assert bench.gather_over_ranks([1.5, 1.0, 2.0], 1) == [[1.5, 1.0, 2.0]]
    assert bench.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and bench.parse_cpulist("") == []


def test_traffic_entry_must_name_the_kernel_and_the_build_that_ran(tmp_path, monkeypatch):
    """roofline.traffic is a committed PMC pass, not a per-run measurement: bench.py quotes it only for the configuration,
    the launch size, the kernel symbol AND the kernel build (hvc_version's id) it has just run"""
    sys.path.insert(0, ROOT)
    import bench
    import video_coding_amd as hvc
    build = hvc.hvc.kernel_build_id()
    assert build == hvc.hvc.kernel_source_id() and len(build) == 12          # the library in the tree is built from the tree's kernels
    assert build.encode() in hvc.lib().hvc_version()
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"kernel": "k_decode_packed", "config": 2, "session": "t", "build": build, "frames_per_launch": 1024, "hbm_bytes": 9.7e9}
    (prof / "traffic.json").write_text(json.dumps({"entries": [entry]}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got, src = bench.measured_traffic(2, 1024)
    assert got == 9700000000 and "k_decode_packed" in src and build in src
    assert bench.measured_traffic(2, 1000)[0] is None                      # another launch size
    got, why = bench.measured_traffic(2, 1024, kernel="k_decode_ldsdma")   # another kernel behind the same step
    assert got is None and "k_decode_packed" in why
    # a pass of another build of the kernels is stale, and says so
    (prof / "traffic.json").write_text(json.dumps({"entries": [dict(entry, build="0123456789ab")]}))
    got, why = bench.measured_traffic(2, 1024)
    assert got is None and why.startswith("stale: profiled build 0123456789ab") and build in why
    # ... a later entry of the running build still counts
    (prof / "traffic.json").write_text(json.dumps({"entries": [dict(entry, build="0123456789ab"), dict(entry, hbm_bytes=5.0)]}))
    assert bench.measured_traffic(2, 1024)[0] == 5
    # an entry without a build id (sessions before round 6) or without a kernel name is not trusted
    (prof / "traffic.json").write_text(json.dumps({"entries": [{k: v for k, v in entry.items() if k != "build"}]}))
    assert bench.measured_traffic(2, 1024)[0] is None
    (prof / "traffic.json").write_text(json.dumps({"entries": [{"config": 2, "frames_per_launch": 1024, "hbm_bytes": 1.0, "build": build}]}))
    assert bench.measured_traffic(2, 1024)[0] is None


def test_committed_traffic_entries_are_of_a_named_build():
    """profiles/traffic.json as committed: every entry of the current session names the kernel build it profiled; what bench.py
    makes of it for the headline is either that pass or an explicit "stale" (never a silent number of another build)"""
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    assert all(len(e.get("build", "")) == 12 for e in t["entries"] if e["session"] >= "r06")   # (round 6 introduced the id)
    got, src = bench.measured_traffic(2, 1024)
    assert (got is not None and "build" in src) or src.startswith("stale: profiled build")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", **extra)
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout  # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_typed_by_hand_launches_its_own_ranks():
    """`python bench.py --gpus 2`, no launcher around it: bench.py starts two fresh ranks itself (child processes
    of torch.distributed.run, before any HIP call), they rendezvous, run W + K steps between barriers, and rank 0
    prints the one JSON line.  HVC_BENCH_NO_GPU=1 replaces the step with a sleep (this machine has no GPU): the
    launch path is the thing under test, and the line says that nothing was decoded."""
    for config, extra in ((2, []), (4, ["--shard", "256"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                              "--config", str(config)] + extra,
                             capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=300, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        rec = _json_line(out.stdout)
        assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1 and rec["scaling"] == "weak"
        assert rec["config"]["step_calls_rank0"] == 5
        assert "not a measurement" in rec["data"]
        # the slow rank (2 ms a step) bounds the job; the value is the whole job's: both ranks' frames
        assert rec["ms_per_step"] >= 2.0
        frames = 1024 if config == 2 else 256
        px = 1920 * 1080 if config == 2 else 3840 * 2160
        assert abs(rec["value"] - 2 * frames * px / (rec["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * rec["value"]


def test_bench_under_the_drivers_launcher_and_alone():
    """the driver's own command line for N > 1 (torch.distributed.run around bench.py) and the plain N = 1 form"""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "0"],
                         capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert _json_line(out.stdout)["n_gpus"] == 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "0"],
                         capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert _json_line(out.stdout)["n_gpus"] == 1
    # a launcher whose world size contradicts --gpus is an error, not a silent re-interpretation
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
                          "--gpus", "4"], capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=300, cwd=ROOT)
    assert out.returncode != 0


def test_eight_ranks_as_the_driver_will_start_them():
    """The scaling run is N = 1, 2, 4, 8 on an 8-GPU node nobody here can reach: the 8-rank launch itself -- self-launch
    as typed and the driver's torch.distributed.run form, configs 2 and 4 -- is rehearsed under gloo (a GPU box admits
    at most 6 processes on its card, so eight ranks can only be rehearsed on the CPU)."""
    for config, extra in ((2, []), (4, ["--shard", "256"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                              "--config", str(config)] + extra,
                             capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1", OMP_NUM_THREADS="1"),
                             timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        rec = _json_line(out.stdout)
        assert rec["n_gpus"] == 8 and rec["steps"] == 3 and rec["scaling"] == "weak"
        assert rec["ms_per_step"] >= 8.0  # rank 7 sleeps 8 ms a step and bounds the job
        # every rank's own figure reaches rank 0's line (gathered over the group that closes the timing): a slow GPU shows
        per_rank = rec["per_rank_kernel_ms"]["mean_min_max"]
        assert [r[0] for r in per_rank] == [float(k + 1) for k in range(8)]
        frames = 1024 if config == 2 else 256
        px = 1920 * 1080 if config == 2 else 3840 * 2160
        assert abs(rec["value"] - 8 * frames * px / (rec["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * rec["value"]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
                          "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1", OMP_NUM_THREADS="1"),
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert _json_line(out.stdout)["n_gpus"] == 8


def test_host_share_splits_a_numa_node_between_the_ranks_on_it():
    """config 3 at N > 1 (host Huffman threads are its scaling limit): ranks whose GPUs hang off one NUMA node
    take equal, disjoint slices of that node's CPUs, 16 threads at most, one at least"""
    sys.path.insert(0, ROOT)
    import bench
    node0, node1 = list(range(0, 64)), list(range(64, 128))
    keys = [0, 0, 0, 0, 64, 64, 64, 64]                      # 8 GPUs, 4 per socket
    got = [bench.host_share(keys, r, node0 if r < 4 else node1) for r in range(8)]
    assert all(t == 16 for _, t in got)
    for node, ranks in ((node0, range(0, 4)), (node1, range(4, 8))):
        seen = [c for r in ranks for c in got[r][0]]
        assert sorted(seen) == node                            # the node is covered, nobody shares a CPU
    assert bench.host_share([0, 0], 1, list(range(16))) == (list(range(8, 16)), 8)
    assert bench.host_share([0] * 8, 7, list(range(4))) == ([3], 1)          # fewer CPUs than ranks: one thread each, the last CPU
    assert bench.host_share([0], 0, list(range(96))) == (list(range(96)), 16)
    assert bench.format_cpulist([5, 0, 1, 2, 8, 9]) == "0-2,5,8-9" and bench.parse_cpulist(bench.format_cpulist(node1)) == node1


def test_config3_and_5_over_n_ranks():
    """`bench.py --config 3` splits BASELINE config 3's 4096 files over the ranks (strong scaling) and gives every rank its share
    of the host's CPUs; `--config 5` is the encoder's block stage, weak.  Launch path only (HVC_BENCH_NO_GPU=1), world 2 and 8,
    typed by hand and in the driver's form."""
    cpus = len(os.sched_getaffinity(0))
    for world in (2, 8):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                              "--config", "3"], capture_output=True, text=True,
                             env=_clean_env(HVC_BENCH_NO_GPU="1", OMP_NUM_THREADS="1"), timeout=600, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-3000:]
        rec = _json_line(out.stdout)
        assert rec["n_gpus"] == world and rec["scaling"] == "strong" and rec["config"]["baseline_config"] == 3
        assert rec["config"]["files_total"] == 4096 and rec["config"]["files_per_gpu_per_step"] == 4096 // world
        assert "4096 x 1080p" in rec["config"]["workload"] and "host Huffman || H2D || K1" in rec["config"]["workload"]
        sys.path.insert(0, ROOT)
        import bench
        quota = bench.cgroup_cpu_quota()
        cap = 16 if quota is None else max(1, min(16, int(quota / world)))
        assert rec["config"]["host_threads_per_rank"] == [max(1, min(cap, cpus // world))] * world
        # the whole job's files over the slowest rank's time
        assert abs(rec["value"] - 4096 * 1920 * 1080 / (rec["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * rec["value"]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29551", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "0", "--config", "5"],
                         capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = _json_line(out.stdout)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["baseline_config"] == 5
    assert abs(rec["value"] - 2 * 256 * 3840 * 2160 / (rec["ms_per_step"] * 1e-3) / 1e6) < 1e-2 * rec["value"]
    # 4096 files do not split over 3 ranks: an error, not a silent remainder
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--config", "3"],
                         capture_output=True, text=True, env=_clean_env(HVC_BENCH_NO_GPU="1"), timeout=120, cwd=ROOT)
    assert out.returncode != 0
