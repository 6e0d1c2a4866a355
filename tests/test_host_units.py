"""The host half of the GPU reader's restart-interval support, without a GPU: hvc::prepare_gpu_decode_to with `units` (every
interval of a file unstuffed into a slot of its own, hvc_hdec.h RstUnits) through tests/host_harness/units_harness.cpp, built
by tests/host_harness/Makefile (g++, CPU only) with AddressSanitizer against csrc/hvc_entropy.cpp.  The intervals it cuts = this file's own cut of the same bytes;
mutated and truncated files never write past the buffers the callers give it."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import jpeg_optimised_tables
from test_restart_intervals import QT, random_record

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS_DIR = os.path.join(ROOT, "tests", "host_harness")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    """tests/host_harness/Makefile: a CPU-only g++ build (no device code) -- the sanitizer flags live there, and this file and
    that directory are in .gpurunignore (the GPU run does not need them)"""
    import shutil
    if not os.path.isdir("/opt/rocm/include") or not shutil.which("make") or not shutil.which("g++"):
        pytest.skip("the harness needs make, g++ and the HIP headers under /opt/rocm/include (hvc_entropy.cpp includes hip_runtime.h)")
    exe = str(tmp_path_factory.mktemp("units") / "units_harness")
    r = subprocess.run(["make", "-s", "-C", HARNESS_DIR, "OUT=" + exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return exe


def fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def cut(jpg, at):
    """the entropy-coded segment from `at` on, cut at its RSTn markers and unstuffed: T.81 B.1.1.2, B.2.1 (a fill 0xFF in front of
    a marker is skipped; any other marker ends the scan)"""
    units, cur, i = [], bytearray(), at
    while i < len(jpg):
        if jpg[i] != 0xFF:
            cur.append(jpg[i])
            i += 1
            continue
        nxt = jpg[i + 1] if i + 1 < len(jpg) else -1
        if nxt == 0x00:
            cur.append(0xFF)
            i += 2
        elif nxt == 0xFF:
            i += 1
        elif 0xD0 <= nxt <= 0xD7:
            units.append(bytes(cur))
            cur = bytearray()
            i += 2
        else:
            break
    units.append(bytes(cur))
    return units


def run(exe, files, tmp_path):
    paths = []
    for k, f in enumerate(files):
        p = tmp_path / ("f%d.jpg" % k)
        p.write_bytes(f)
        paths.append(str(p))
    r = subprocess.run([exe] + paths, capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1"})
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    return [ln.split()[1:] for ln in r.stdout.splitlines()]


@pytest.mark.parametrize("sampling", [[(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 1)]])
def test_intervals_are_cut_as_the_markers_say(harness, tmp_path, sampling):
    import video_coding_amd as hvc
    files, want = [], []
    for (w, h) in ((64, 48), (200, 72)):
        rec, n_mcu = random_record(sampling, w, h, 3 * w + len(sampling))
        for ri in (1, 2, 7, n_mcu - 1, n_mcu, n_mcu + 3):
            f = jpeg_optimised_tables(w, h, sampling, QT, rec, table_sets=min(2, len(sampling)), restart_interval=ri)
            files.append(f)
            want.append((ri, n_mcu, cut(f, hvc.hvc.jpeg_read_header(f).ecs_offset)))
    for line, (ri, n_mcu, units) in zip(run(harness, files, tmp_path), want):
        if ri >= n_mcu:   # one interval: the plain segment
            assert line[0] == "PLAIN" and line[1] == "ok=1" and line[2] == "bytes=%d" % len(units[0]), line[:4]
            continue
        ipf = -(-n_mcu // ri)
        assert len(units) == ipf
        assert line[:5] == ["UNITS", "ok=1", "ri=%d" % ri, "ipf=%d" % ipf, "bytes=%d" % sum(len(u) for u in units)], line[:5]
        got = line[5:]
        assert len(got) == 2 * ipf
        for k, u in enumerate(units):
            assert int(got[2 * k]) == len(u) and int(got[2 * k + 1], 16) == fnv(u), (ri, k)


def test_mutated_files_stay_inside_the_buffers(harness, tmp_path):
    """markers removed, doubled, moved, bytes changed, files cut: ok=0 (the host reader's file) or a layout that passes the
    harness's own checks -- and no report from AddressSanitizer (halt_on_error: the run would fail)"""
    rng = np.random.Generator(np.random.PCG64(17))
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], 200, 72, 5)
    seeds = [jpeg_optimised_tables(200, 72, 420, QT, rec, restart_interval=ri) for ri in (1, 5, 13)]
    files = []
    for it in range(300):
        b = bytearray(seeds[it % 3])
        kind = it % 5
        marks = [i for i in range(len(b) - 1) if b[i] == 0xFF and 0xD0 <= b[i + 1] <= 0xD7]
        m = marks[int(rng.integers(0, len(marks)))]
        if kind == 0:
            b = b[:int(rng.integers(2, len(b)))]
        elif kind == 1:
            b = b[:m] + b[m + 2:]
        elif kind == 2:
            b = b[:m] + b[m:m + 2] * int(rng.integers(2, 40)) + b[m:]
        elif kind == 3:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            b[int(rng.integers(20, len(b)))] = 0xFF
        files.append(bytes(b))
    lines = run(harness, files, tmp_path)
    assert len(lines) == len(files)
    kinds = {ln[0] for ln in lines}
    assert "UNITS" in kinds and any(ln[0] == "UNITS" and ln[1] == "ok=0" for ln in lines) and any(ln[0] == "UNITS" and ln[1] == "ok=1" for ln in lines)
