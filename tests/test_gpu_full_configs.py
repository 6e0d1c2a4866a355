"""BASELINE.json's configurations at their FULL sizes, every output record K5-checksummed on the device against the golden
values the CPU suite derives from the model restatement (tests/golden/bench_checksums.json, tests/test_bench_checksums.py):
config 3 at 4096 x 1080p files with the host reader and with the GPU reader, config 4 as one GPU's whole 2048-frame 4K 4:4:4
shard (153 GB resident; `bench.py --config 4`), config 5 at 256 x 4K 4:2:0 frames -- block stage and both file pipelines.
(Smaller cases against the oracle itself: test_gpu_fullsize_pipeline.py, test_gpu_fullsize_properties.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_config3_at_4096_files_both_readers():
    import bench_configs as bc
    for gpu in (False, True):
        r = bc.config3(bc.make_args(frames=4096, steps=1, threads=16, chunk=256 if gpu else 32, gpu_entropy=gpu))
        assert r["frames"] == 4096 and r["checksum"]["records"] == 4096 and r["checksum"]["verified"] is True, r["config"]
        assert r["value"] > 1000.0        # (Mpixel/s: a pipeline that has fallen off a cliff would still verify)


def test_config5_at_256_frames_block_stage_and_files():
    import bench_configs as bc
    r = bc.config5(bc.make_args(frames=256, steps=2, warmup=1))
    assert r["checksum"]["records"] == 256 and r["checksum"]["verified"] is True
    for gpu in (False, True):
        r = bc.config5_files(bc.make_args(frames=256, steps=1, threads=16, chunk=16, gpu_entropy=gpu))
        assert r["checksum"]["records"] == 256 and r["checksum"]["verified"] is True, r["config"]


def test_config4_one_gpus_whole_shard():
    """2048 frames of 4K 4:4:4 per GPU, resident when the device has the room (the line says which), 16 launches of 128 frames
    per step: `bench.py --config 4` exits non-zero unless every distinct frame's checksum is the model's"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "4", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--sustain-seconds", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["checksum"]["verified"] is True and rec["config"]["frames_per_gpu_per_step"] == 2048
    assert rec["config"]["frames_per_launch"] == 128 and rec["config"]["baseline_config"] == 4
    assert "resident" in rec["config"]["workload"]


def test_convert_passes_on_resident_frames():
    """tools/bench_configs.py --config 11: whole `oyuv convert` passes (4:2:0 <-> 4:4:4, to and from packed 4:2:2, a crop at an
    offset) on 1080p frames resident in HBM -- every output frame's K5 checksum is the restated Oconv's"""
    import bench_configs as bc
    r = bc.config_convert(bc.make_args(frames=12, steps=1, warmup=0))
    assert [p["pass"] for p in r["passes"]] == [k for k, *_ in bc.CONVERT_PASSES]
    assert all(p["verified"] is True for p in r["passes"]) and r["checksum"]["verified"] is True
