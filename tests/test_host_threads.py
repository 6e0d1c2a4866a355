"""The host threads of the batch pipelines (csrc/hvc_pool.h): a persistent pool inside the context, and no C++
exception across the C boundary.  The model itself is single-threaded; these threads are the
library's own, so what is pinned here is the ABI's promise (include/hvc_jpeg.h "no exceptions cross the boundary"):
a thread the system refuses to start is HVC_E_SYSTEM, not std::terminate.  No GPU needed: hvc_host_threads_probe
builds a pool the way a batch call does."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HVC_E_SYSTEM = -10

CHILD = r'''
import ctypes, os, resource, sys
sys.path.insert(0, %r)
mode = sys.argv[1]
import video_coding_amd as hvc
L = hvc.lib()                       # (loaded before any limit: the loader itself needs no thread)
assert L.hvc_host_threads_probe(4) == 0
if mode == "rlimit":
    soft, hard = resource.getrlimit(resource.RLIMIT_NPROC)
    resource.setrlimit(resource.RLIMIT_NPROC, (1, hard))   # this user already runs more than one task
r = L.hvc_host_threads_probe(int(sys.argv[2]))
print("probe", r, L.hvc_strerror(r).decode())
r2 = L.hvc_host_threads_probe(0)
print("bad", r2)
''' % ROOT


def _child(mode, n, **env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD, mode, str(n)], capture_output=True, text=True, env=e, timeout=300, cwd=ROOT)
    # an abort (std::terminate) would show as a negative return code and no "probe" line
    assert out.returncode == 0, (out.returncode, out.stdout[-1000:], out.stderr[-2000:])
    return dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln)


def test_probe_starts_and_joins_threads():
    got = _child("plain", 16)
    assert got["probe"].startswith("0 ") and got["bad"] == "-1"


def test_refused_thread_is_an_error_code_not_an_abort():
    """HVC_POOL_FAIL_AFTER=k makes the k-th thread creation (and every later one) throw the std::system_error a pids
    limit produces, with k - 1 threads already running -- the case that ended in std::terminate before."""
    got = _child("inject", 8, HVC_POOL_FAIL_AFTER="6")  # 4 threads of the first probe + 2 of the second, then EAGAIN
    assert got["probe"].startswith("%d " % HVC_E_SYSTEM), got
    assert "thread" in got["probe"]


@pytest.mark.skipif(os.geteuid() == 0, reason="RLIMIT_NPROC is not enforced for root")
def test_refused_thread_under_rlimit_nproc():
    got = _child("rlimit", 8)
    assert got["probe"].startswith("%d " % HVC_E_SYSTEM), got


def test_every_entry_point_is_a_function_try_block():
    """include/hvc_jpeg.h: "no exceptions cross the boundary".  Every int-returning entry point defined in csrc/ is a
    function-try-block ending in HVC_ABI_CATCH (one-line forwarders to another entry point excepted)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "hvc_jpeg.h")).read()
    declared = set(re.findall(r"HVC_API\s+int\s+(hvc_\w+)\s*\(", hdr))
    guarded, forwarders = set(), set()
    for fn in ("hvc_capi.hip", "hvc_capi_jpeg.hip", "hvc_capi_reader.hip", "hvc_capi_files.hip", "hvc_capi_async.hip", "hvc_yuv.hip", "hvc_entropy.cpp"):
        txt = open(os.path.join(ROOT, "video-coding_amd", "csrc", fn)).read()
        for m in re.finditer(r"^int (hvc_\w+)\(([^{;]*?)\)\s*(try\s*)?\{([^\n]*)", txt, re.M | re.S):
            name, is_try, rest = m.group(1), m.group(3), m.group(4)
            if is_try:
                guarded.add(name)
            elif rest.rstrip().endswith("}"):
                forwarders.add(name)
        assert txt.count("HVC_ABI_CATCH") == len(re.findall(r"\)\s*try\s*\{", txt)) or fn.startswith("hvc_capi")
    assert declared - guarded - forwarders == set(), declared - guarded - forwarders
    assert forwarders <= {"hvc_last_hip_error", "hvc_last_kernel_ms"}
