"""K5 (hvc_checksum_records) against its three-line numpy definition, the benchmarks' verified output, the
N-rank launch path of bench.py on the GPU box, and contexts opened by device index."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN
from helpers import checksum_records, synth_coefs
from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,size,stride", [(1, 1, 1), (3, 15, 15), (5, 16, 16), (4, 1000, 1024), (2, 4097, 4099),
                                           (7, 3133440, 3133440), (1, 40_000_003, 40_000_003), (300, 64, 64)])
def test_checksum_kernel_equals_its_definition(ctx, n, size, stride):
    import torch
    rng = np.random.Generator(np.random.PCG64(n * 131 + size))
    buf = rng.integers(0, 256, size=(n - 1) * stride + size, dtype=np.uint8)
    recs = np.stack([buf[r * stride:r * stride + size] for r in range(n)])
    want = checksum_records(recs)
    assert np.array_equal(ctx.checksum_records(buf, size, n, stride), want)            # host data
    d = torch.from_numpy(buf).cuda()
    torch.cuda.synchronize()
    assert np.array_equal(ctx.checksum_records(d, size, n, stride), want)              # device data
    if size > 40:  # unaligned records take the byte path
        assert np.array_equal(ctx.checksum_records(d[3:], size - 3, 1, stride), checksum_records(recs[:1, 3:]))
    assert ctx.checksum_records(d, size, 0).size == 0


def test_checksum_of_decoded_frames_is_the_models(ctx):
    """the benchmarks' chain in small: coefficients -> hvc_decode_frames (device) -> K5 == checksum of the
    model restatement's planes"""
    import torch
    import video_coding_amd as hvc
    q = orc.quant_scale(orc.quant_luma(), 75).astype(np.uint16)
    bw, bh, n = 40, 23, 6
    coefs = np.stack([synth_coefs(300 + i, bh, bw, q)[0] for i in range(n)])
    want = checksum_records(orc.dequant_idct_recon(coefs, q, bw, bh, n_planes=n).reshape(n, -1))
    d_c = torch.from_numpy(coefs).cuda()
    d_p = torch.zeros((n, bh * 8 * bw * 8), dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.dequant_idct_recon(d_c, q, bw, bh, n, d_p)
        assert np.array_equal(ctx.checksum_records(d_p, bh * 8 * bw * 8, n), want)
    finally:
        ctx.reset_stream()


def _run_bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_is_verified_against_the_golden_checksums():
    """python bench.py (config 2, small batch): the line carries roofline + cpu_baseline + a checksum that equals
    tests/golden/bench_checksums.json -- entries the CPU suite derives from the model restatement."""
    rec = _run_bench(["--steps", "3", "--warmup", "1", "--frames", "64", "--cpu-seconds", "1", "--sustain-seconds", "0.5"])
    with open(os.path.join(GOLDEN, "bench_checksums.json")) as f:
        g = json.load(f)
    assert rec["checksum"]["verified"] is True and rec["checksum"]["rank0"] == g["bench_config2"]["rank0"]
    assert rec["n_gpus"] == 1 and rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == 1
    assert rec["timed_region_s"] > 0 and rec["roofline"]["traffic_source"]
    assert rec["config"]["wide_path_blocks"] == 0
    su = rec["sustained"]  # the same step repeated after the timed region, never part of value
    assert su["wall_s"] >= 0.5 and su["steps"] >= 64 and su["output_unchanged"] is True and 0 < su["first_decile_ms"] < 50


def test_headline_launch_verifies_every_one_of_its_1024_frames():
    """`python bench.py` at its default launch -- 1024 frames of 1080p 4:2:0 -- K5-checksums EVERY record of the
    launch whose time is `value` (record r holds distinct frame r % 8), not the first eight; the tight layout beside the aligned
    one is timed and verified in the same run (roofline.other_layout)."""
    rec = _run_bench(["--steps", "2", "--no-others", "--cpu-seconds", "0", "--sustain-seconds", "0.3"])
    assert rec["config"]["frames_per_launch"] == 1024 and rec["config"]["baseline_config"] == 2
    assert rec["checksum"]["frames"] == 1024 and rec["checksum"]["distinct"] == 8 and rec["checksum"]["verified"] is True
    assert rec["sustained"]["output_unchanged"] is True
    o = rec["roofline"]["other_layout"]
    assert "tight" in o["layout"] and o["verified"] is True and 0.3 < o["frac"] < 1.0
    assert rec["roofline"]["traffic"] or rec["roofline"]["traffic_source"].startswith(("stale", "no committed"))


def test_bench_layouts_aligned_and_tight_decode_the_same_frames():
    """bench.py lays its resident batch out with every plane on a 64 KiB / 2 MiB boundary (hvc.layout_alignment); --tight = planes back
    to back.  Both verify against the same golden checksums (K5 runs on the planes gathered tight), configs 2 and 5."""
    for extra in ([], ["--config", "5"]):
        recs = [_run_bench(extra + ["--steps", "2", "--warmup", "1", "--frames", "32", "--no-cpu-baseline", "--no-others", "--sustain-seconds", "0"] + t)
                for t in ([], ["--tight"])]
        assert "boundary" in recs[0]["config"]["layout"] and "tight" in recs[1]["config"]["layout"]
        assert all(r["checksum"]["verified"] is True for r in recs)
        assert recs[0]["checksum"]["rank0"] == recs[1]["checksum"]["rank0"]


def test_bench_n_rank_path_on_one_gpu():
    """HVC_BENCH_REHEARSAL=1 python bench.py --gpus 2, as typed: bench.py starts the two ranks itself (both on
    cuda:0, gloo for the timing closure), each decodes its own shard, both outputs are verified."""
    rec = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "64"], HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 2 and "REHEARSAL" in rec["data"]
    assert rec["checksum"]["verified"] is True and rec["checksum"]["ranks_verified"] == 2
    rec = _run_bench(["--gpus", "2", "--config", "4", "--steps", "1", "--warmup", "0", "--frames", "8", "--shard", "16"],
                     HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 2 and rec["config"]["baseline_config"] == 4 and rec["checksum"]["ranks_verified"] == 2
    assert rec["config"]["frames_per_gpu_per_step"] == 16 and rec["config"]["frames_per_launch"] == 8


def test_bench_four_ranks_on_one_gpu():
    """The widest rehearsal a one-GPU box allows (at most 6 processes may use its card; eight ranks are rehearsed on
    the CPU, tests/test_distributed_cpu.py): four ranks of both configurations, every rank's shard verified against
    the golden checksums of ITS rank (distinct seeds per rank)."""
    rec = _run_bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--frames", "32", "--sustain-seconds", "0.2"],
                     HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 4 and rec["checksum"]["ranks_verified"] == 4 and "REHEARSAL" in rec["data"]
    assert rec["sustained"]["output_unchanged"] is True
    rec = _run_bench(["--gpus", "4", "--config", "4", "--steps", "1", "--warmup", "0", "--frames", "8", "--shard", "16",
                      "--sustain-seconds", "0"], HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 4 and rec["config"]["baseline_config"] == 4 and rec["checksum"]["ranks_verified"] == 4
    assert "sustained" not in rec


def test_bench_config3_and_5_one_rank_and_two():
    """bench.py --config 3 (BASELINE configs[2]: JPEG files, host Huffman || H2D || K1; the GPU reader beside it) and --config 5
    (configs[4]: the encoder's block stage), small batches: alone with their CPU baselines, and over two ranks sharing the GPU
    (the file batch split, every rank's output K5-verified, each rank on its share of the host's CPUs)."""
    rec = _run_bench(["--config", "3", "--frames", "128", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1"])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["config"]["baseline_config"] == 3
    assert "host Huffman || H2D || K1" in rec["config"]["workload"] and rec["config"]["files_total"] == 128
    for reader in ("host_reader", "gpu_reader"):
        assert rec[reader]["verified"] is True and rec[reader]["ranks_verified"] == 1 and rec[reader]["value"] > 0
    assert rec["value"] == rec["host_reader"]["value"] and rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == 1
    # upload rate, overlap fraction and the counter bytes of the pipeline's own launches
    assert rec["host_reader"]["h2d_GBps"] > 1 and -1.0 < rec["host_reader"]["overlap_fraction"] < 1.0
    rf = rec["roofline"]
    assert rf["h2d_GBps"] == rec["host_reader"]["h2d_GBps"] and rf["overlap_fraction"] == rec["host_reader"]["overlap_fraction"]
    assert rf["algorithmic_bytes_per_launch"] == rec["host_reader"]["kernel_launch_frames"] * 48960 * 192
    assert rf["traffic"] or rf["traffic_source"].startswith(("stale", "no committed"))
    rec = _run_bench(["--gpus", "2", "--config", "3", "--frames", "128", "--steps", "2", "--warmup", "1"], HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 2 and rec["config"]["files_per_gpu_per_step"] == 64 and "REHEARSAL" in rec["data"]
    assert rec["host_reader"]["ranks_verified"] == 2 and rec["gpu_reader"]["ranks_verified"] == 2
    walls = rec["host_reader"]["per_rank"]["wall_ms_entropy_thread_ms_sum_h2d_ms_kernel_ms_threads"]
    assert len(walls) == 2 and all(w[0] > 0 and w[4] >= 1 for w in walls)
    rec = _run_bench(["--config", "5", "--frames", "16", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1"])
    assert rec["config"]["baseline_config"] == 5 and rec["roofline"]["kernel"] == "k_encode" and rec["checksum"]["verified"] is True
    assert rec["cpu_baseline"]["value"] > 0 and 0 < rec["roofline"]["frac"] < 1
    rec = _run_bench(["--gpus", "2", "--config", "5", "--frames", "16", "--steps", "2", "--warmup", "1"], HVC_BENCH_REHEARSAL="1")
    assert rec["n_gpus"] == 2 and rec["checksum"]["ranks_verified"] == 2 and len(rec["per_rank_kernel_ms"]["mean_min_max"]) == 2


def test_bench_fails_when_the_output_is_not_the_models(tmp_path):
    """A run whose checksums differ from the golden ones exits non-zero and says "verified": false: here
    the golden file is swapped for one with another rank's values."""
    with open(os.path.join(GOLDEN, "bench_checksums.json")) as f:
        g = json.load(f)
    g["bench_config2"]["rank0"] = g["bench_config2"]["rank1"]
    alt = tmp_path / "bench_checksums.json"
    alt.write_text(json.dumps(g))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e["HVC_BENCH_GOLDEN"] = str(alt)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "32",
                          "--no-cpu-baseline", "--sustain-seconds", "0"], capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert out.returncode == 3, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["checksum"]["verified"] is False and rec["checksum"]["ranks_verified"] == 0
    # more distinct frames than the golden file holds: the line says the comparison was skipped, exit 0
    rec = _run_bench(["--steps", "2", "--warmup", "1", "--frames", "32", "--distinct", "9", "--no-cpu-baseline", "--sustain-seconds", "0"])
    assert rec["checksum"]["verified"] is None and rec["checksum"]["verification"].startswith("SKIPPED")


def test_contexts_by_device_index():
    """hvc_create(device): one context per visible GPU (whatever their number on this machine: 1 on a one-GPU
    box, 8 on a node), each decoding on ITS device; an index past the last one is HVC_E_NO_DEVICE."""
    import torch
    import video_coding_amd as hvc
    n_dev = torch.cuda.device_count()
    assert n_dev >= 1
    q = orc.quant_scale(orc.quant_chroma(), 60).astype(np.uint16)
    bw, bh = 33, 9
    coefs, _ = synth_coefs(4242, bh, bw, q)
    want = orc.dequant_idct_recon(coefs, q, bw, bh).reshape(bh * 8, bw * 8)
    ctxs = [hvc.Context(dev) for dev in range(n_dev)] + [hvc.Context(n_dev - 1)]  # and a second one on the last device
    try:
        for i, c in enumerate(ctxs):
            dev = min(i, n_dev - 1)
            with torch.cuda.device(dev):
                d_c = torch.from_numpy(coefs).to("cuda:%d" % dev)
                d_p = torch.zeros((bh * 8, bw * 8), dtype=torch.uint8, device="cuda:%d" % dev)
                torch.cuda.synchronize(dev)
            c.dequant_idct_recon(d_c, q, bw, bh, 1, d_p)  # called with ANOTHER device current: the context switches itself
            c.synchronize()
            assert np.array_equal(d_p.cpu().numpy(), want), dev
            out = np.zeros((bh * 8, bw * 8), dtype=np.uint8)
            c.dequant_idct_recon(coefs, q, bw, bh, 1, out)
            assert np.array_equal(out, want), dev
    finally:
        for c in ctxs:
            c.close()
    for bad in (n_dev, n_dev + 7, -1):
        with pytest.raises(hvc.HvcError) as e:
            hvc.Context(bad)
        assert e.value.code == -2


def test_bench_rccl_calls_on_one_gpu():
    """The N > 1 path of bench.py closes its timing with RCCL (backend "nccl"): init_process_group with the rank's
    device, barriers, a MAX all-reduce and the SUM all-reduce of the verification flags.  A one-GPU box cannot hold two
    RCCL ranks, but it can run exactly those calls in a world of one: the driver's launcher around bench.py with
    one process and HVC_BENCH_DIST_ALWAYS=1."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(HVC_BENCH_DIST_ALWAYS="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29551", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--frames", "64", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["checksum"]["ranks_verified"] == 1 and "REHEARSAL" not in rec["data"]
    # ... and the same calls (plus the all-gather of the per-rank figures) in the file pipeline's and the encoder's N-rank paths
    for port, extra in (("29553", ["--config", "3", "--frames", "64", "--steps", "1", "--warmup", "1"]),
                        ("29555", ["--config", "5", "--frames", "16", "--steps", "2", "--warmup", "1"])):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                              "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
                              "--gpus", "1", "--no-cpu-baseline"] + extra, capture_output=True, text=True, env=e, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert rec["n_gpus"] == 1 and rec["checksum"]["ranks_verified"] == 1 and "REHEARSAL" not in rec["data"]
