"""tests/golden/bench_checksums.json pins what the benchmarks decode: bench.py and tools/bench_configs.py checksum
their output on the device (K5, hvc_checksum_records) and compare with this file; here the CPU restatement of the
model re-derives entries of it from the benchmarks' seeds (a spot check per configuration -- the whole file is
what tests/golden/make_bench_checksums.py writes, 3 minutes of CPU).  So a `verified: true` in a bench line
means: the timed output equals the model's output on the same inputs."""
import importlib.util
import json
import os

import numpy as np

from conftest import GOLDEN
from helpers import checksum_records

spec = importlib.util.spec_from_file_location("make_bench_checksums", os.path.join(GOLDEN, "make_bench_checksums.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)

with open(os.path.join(GOLDEN, "bench_checksums.json")) as f:
    G = json.load(f)


def test_checksum_definition():
    """sum_i (b_i + 1) (2 i + 1) M mod 2^64: position-weighted, order-free, never blind to a changed byte"""
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.integers(0, 256, size=(3, 1000), dtype=np.uint8)
    M = 0x9E3779B97F4A7C15
    want = [sum((int(b) + 1) * (2 * i + 1) * M for i, b in enumerate(row)) % (1 << 64) for row in a]
    assert [int(x) for x in checksum_records(a)] == want
    b = a.copy()
    b[1, 777] ^= 1
    assert checksum_records(b)[1] != checksum_records(a)[1] and checksum_records(b)[0] == checksum_records(a)[0]
    z = np.zeros((1, 64), np.uint8)  # all-zero records of different lengths differ (the + 1)
    assert checksum_records(z)[0] != checksum_records(z[:, :63])[0]
    # any split into parts sums to the whole (what lets the device add partial sums in any order)
    w = (np.arange(1000, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) * np.uint64(M)
    with np.errstate(over="ignore"):
        parts = sum(int(((a[0, s:s + 100].astype(np.uint64) + np.uint64(1)) * w[s:s + 100]).sum(dtype=np.uint64)) for s in range(0, 1000, 100))
    assert parts % (1 << 64) == want[0]


def test_bench_config2_entries_follow_from_the_model():
    for rank, frames in ((0, 2), (5, 1)):
        got = ["%016x" % int(checksum_records(rec[None, :])[0]) for rec in mk.bench_frames(2, rank, frames)]
        assert got == G["bench_config2"]["rank%d" % rank][:frames]
    assert all(len(G["bench_config2"]["rank%d" % r]) == 8 for r in range(8))


def test_bench_config4_entry_follows_from_the_model():
    got = ["%016x" % int(checksum_records(rec[None, :])[0]) for rec in mk.bench_frames(4, 0, 1)]
    assert got == G["bench_config4"]["rank0"][:1]
    assert all(len(G["bench_config4"]["rank%d" % r]) == 8 for r in range(8))


def test_tools_bench_configs_entries_follow_from_the_model():
    assert mk.c3_entry(1) == G["configs_c3"][:1]
    assert mk.c5_entry(1) == G["configs_c5"][:1]
    assert mk.c7_entry(1) == G["configs_c7"][:1]
    assert mk.k2_entry(1) == G["configs_k2"][:1]
    assert mk.sub420_entry(1) == G["configs_sub420"][:1]
    assert mk.c5_files_entry(1) == G["configs_c5_files"][:1]
    for key, sums in mk.convert_entries(1, only=("420_to_uyvy", "420_crop_720p")).items():
        assert sums == G[key][:1]
    assert sum(k.startswith("configs_convert_") for k in G) == 5
