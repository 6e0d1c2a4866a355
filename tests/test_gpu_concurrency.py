"""Several contexts at once, as on a multi-GPU node with one host thread (or process) per GPU: two contexts run
their BATCH pipelines concurrently -- host worker threads, pinned rings, copy / reader / block-stage streams each --
and the host threads stay on the CPUs hvc_set_host_cpus gives them."""
import glob
import os
import threading
import time

import numpy as np
import pytest

from helpers import synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu


def _frames(n, w, h, seed):
    out = []
    for f in range(n):
        y = synth_pixels(seed + f, h, w)
        u = synth_pixels(seed + 100 + f, h // 2, w // 2)
        v = synth_pixels(seed + 200 + f, h // 2, w // 2)
        out.append((y, u, v))
    return out


def test_two_contexts_run_their_batch_pipelines_concurrently():
    """context A decodes a batch of files (GPU Huffman reader, then the host-reader pipeline), context B encodes raw
    frames to files (GPU coder, then host coder) at the same time, several rounds; every result against the model"""
    import video_coding_amd as hvc
    w, h, n = 640, 352, 24
    frames_a, frames_b = _frames(n, w, h, 5000), _frames(n, w, h, 7000)
    jpegs_a = [orc.encode_yuv(y, u, v, w, h, 420, 75) for y, u, v in frames_a]
    want_b = [orc.encode_yuv(y, u, v, w, h, 420, 60) for y, u, v in frames_b]
    want_a = []
    for j in jpegs_a:
        d = orc.Decoder(j)
        d.decode()
        want_a.append(np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
    raw_b = [np.concatenate([p.reshape(-1) for p in f]) for f in frames_b]
    a, b = hvc.Context(0), hvc.Context(0)
    info = hvc.hvc.jpeg_read_header(jpegs_a[0])
    fs = info.pixel_bytes
    errors = []

    def run_a():
        try:
            for rep in range(4):
                out = np.zeros(n * fs, np.uint8)
                a.jpeg_decode_batch(jpegs_a, out, fs, threads=4, frames_per_chunk=5, gpu_entropy=rep % 2 == 0)
                for f in range(n):
                    assert np.array_equal(out[f * fs:(f + 1) * fs], want_a[f]), ("decode", rep, f)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def run_b():
        try:
            for rep in range(4):
                got, _ = b.jpeg_encode_batch(raw_b, w, h, 420, 60, threads=4, frames_per_chunk=5, gpu_entropy=rep % 2 == 0)
                assert got == want_b, ("encode", rep)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ta, tb = threading.Thread(target=run_a), threading.Thread(target=run_b)
    ta.start()
    tb.start()
    ta.join()
    tb.join()
    a.close()
    b.close()
    assert not errors, errors


def _thread_cpu_lists():
    out = []
    for p in glob.glob("/proc/self/task/*/status"):
        try:
            with open(p) as f:
                for line in f:
                    if line.startswith("Cpus_allowed_list:"):
                        out.append(line.split(":")[1].strip())
        except OSError:
            pass
    return out


def test_host_threads_stay_on_the_cpus_they_are_given():
    import video_coding_amd as hvc
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 3:
        pytest.skip("needs three CPUs to tell a restricted thread from an unrestricted one")
    c = hvc.Context(0)
    try:
        assert c.get_host_cpus() == ("", 0) or os.environ.get("HVC_HOST_CPUS")
        for bad in ("3-1", "abc", "1,,2", "-4", "999999"):
            with pytest.raises(hvc.HvcError) as e:
                c.set_host_cpus(bad)
            assert e.value.code == -1
        pair = "%d,%d" % (allowed[0], allowed[1])
        c.set_host_cpus(pair)
        assert c.get_host_cpus() == (pair, 2)
        # a batch long enough to look at its threads while it runs
        w, h, n = 640, 352, 96
        j = [orc.encode_yuv(*f, w, h, 420, 75) for f in _frames(4, w, h, 9000)]
        jpegs = [j[i % 4] for i in range(n)]
        info = hvc.hvc.jpeg_read_header(jpegs[0])
        fs = info.pixel_bytes
        out = np.zeros(n * fs, np.uint8)
        seen, stop = set(), threading.Event()

        def watch():
            while not stop.is_set():
                seen.update(_thread_cpu_lists())
                time.sleep(0.0005)

        t = threading.Thread(target=watch)
        t.start()
        for gpu in (False, True, False):
            c.jpeg_decode_batch(jpegs, out, fs, threads=6, frames_per_chunk=8, gpu_entropy=gpu)
        stop.set()
        t.join()
        want_list = pair if allowed[1] != allowed[0] + 1 else "%d-%d" % (allowed[0], allowed[1])
        assert want_list in seen, seen  # worker threads restricted to the pair were observed
        d = orc.Decoder(jpegs[0])
        d.decode()
        assert np.array_equal(out[:fs], np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
        # "auto": the CPUs of the GPU's NUMA node (when sysfs shows them in this container)
        try:
            c.set_host_cpus("auto")
            lst, k = c.get_host_cpus()
            assert k >= 1 and lst
        except hvc.HvcError as e:
            assert e.code == -1
        c.set_host_cpus(None)
        assert c.get_host_cpus() == ("", 0)
    finally:
        c.close()


def test_batch_calls_reuse_the_contexts_threads():
    """The workers live in the context (csrc/hvc_pool.h): a loop of small batch calls starts them once.  Every kind of
    batch call, host and device side of the pool (the downloader of host output is one more thread)."""
    import video_coding_amd as hvc
    w, h, n = 320, 176, 12
    frames = _frames(n, w, h, 11000)
    jpegs = [orc.encode_yuv(y, u, v, w, h, 420, 75) for y, u, v in frames]
    raw = [np.concatenate([p.reshape(-1) for p in f]) for f in frames]
    want = []
    for j in jpegs:
        d = orc.Decoder(j)
        d.decode()
        want.append(np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
    c = hvc.Context(0)
    try:
        assert c.host_threads() == (0, 0)  # nothing until a batch call needs them
        fs = hvc.hvc.jpeg_read_header(jpegs[0]).pixel_bytes
        for rep in range(6):
            out = np.zeros(n * fs, np.uint8)
            c.jpeg_decode_batch(jpegs, out, fs, threads=5, frames_per_chunk=4, gpu_entropy=rep % 2 == 1)
            for f in range(n):
                assert np.array_equal(out[f * fs:(f + 1) * fs], want[f]), (rep, f)
            got, _ = c.jpeg_encode_batch(raw, w, h, 420, 75, threads=3, frames_per_chunk=4, gpu_entropy=rep % 2 == 0)
            assert got == jpegs, rep
        alive, ever = c.host_threads()
        assert alive == ever == 6, (alive, ever)  # 5 workers + the host-output downloader, started once in 12 calls
        c.jpeg_decode_batch(jpegs, out, fs, threads=9, frames_per_chunk=4, gpu_entropy=True)  # a wider call adds threads
        assert c.host_threads() == (10, 10)
        c.jpeg_decode_batch(jpegs, out, fs, threads=2, frames_per_chunk=4, gpu_entropy=False)  # a narrower one none
        assert c.host_threads() == (10, 10)
    finally:
        c.close()


def test_a_refused_host_thread_fails_the_call_not_the_process():
    """EAGAIN from thread creation (a pids limit: 8 ranks x 16 workers per node) inside a real batch call, with some
    workers already running: HVC_E_SYSTEM, and the same context then serves a call that fits the threads it has."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, "tests")
import numpy as np
from helpers import synth_pixels
from oracle import orc
import video_coding_amd as hvc
w, h, n = 320, 176, 8
fr = [(synth_pixels(1 + f, h, w), synth_pixels(50 + f, h // 2, w // 2), synth_pixels(90 + f, h // 2, w // 2)) for f in range(n)]
jpegs = [orc.encode_yuv(y, u, v, w, h, 420, 75) for y, u, v in fr]
raw = [np.concatenate([p.reshape(-1) for p in f]) for f in fr]
c = hvc.Context(0)
fs = hvc.hvc.jpeg_read_header(jpegs[0]).pixel_bytes
out = np.zeros(n * fs, np.uint8)
for gpu in (False, True):
    try:
        c.jpeg_decode_batch(jpegs, out, fs, threads=8, frames_per_chunk=3, gpu_entropy=gpu)
        raise SystemExit("the call should have failed")
    except hvc.HvcError as e:
        assert e.code == -10, e
try:
    c.jpeg_encode_batch(raw, w, h, 420, 75, threads=8, frames_per_chunk=3)
    raise SystemExit("the call should have failed")
except hvc.HvcError as e:
    assert e.code == -10, e
assert c.host_threads() == (3, 3)          # the three that did start stay in the pool
c.jpeg_decode_batch(jpegs, out, fs, threads=2, frames_per_chunk=3, gpu_entropy=True)   # 2 workers + the downloader
d = orc.Decoder(jpegs[5]); d.decode()
assert np.array_equal(out[5 * fs:6 * fs], np.concatenate([d.plane(i).reshape(-1) for i in range(3)]))
got, _ = c.jpeg_encode_batch(raw, w, h, 420, 75, threads=3, frames_per_chunk=3, gpu_entropy=True)
assert got == jpegs
c.close()
print("refusal ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HVC_POOL_FAIL_AFTER="3")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "refusal ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_every_context_entry_point_refuses_null_buffers():
    """The same sweep as tests/test_abi_symbols.py with a live context in front: every function that takes the context
    called with it and otherwise nothing but zeros and null pointers -- a status comes back (never a positive one), the
    process is still there, and the context then decodes a file like before.  In a child process."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", "hvc_jpeg.h")).read(), flags=re.S)
    protos = re.findall(r"HVC_API\s+int\s+(hvc_\w+)\s*\(\s*(?:const\s+)?hvc_ctx\s*\*\s*\w+\s*((?:,[^;]*?)?)\)\s*;", hdr, flags=re.S)
    calls = [(name, len(re.split(r",(?![^()]*\))", rest)) - 1 if rest.strip() else 0) for name, rest in protos]
    assert len(calls) >= 30
    code = "\n".join([
        "import ctypes as C, sys, numpy as np",
        "sys.path.insert(0, %r); sys.path.insert(0, %r)" % (root, os.path.join(root, "tests")),
        "import video_coding_amd as hvc",
        "from oracle import orc",
        "ctx = hvc.Context(0); L = hvc.lib()",
        "for name, n in %r:" % (calls,),
        "    f = getattr(L, name); f.restype = C.c_int; f.argtypes = [C.c_void_p] * (n + 1)",
        "    print(name, f(ctx._h, *([None] * n)), flush=True)",
        "data = open(%r, 'rb').read()" % os.path.join(root, "tests", "golden", "Mouse480.jpg"),
        "info, pixels = ctx.jpeg_decode(data)",
        "d = orc.Decoder(data); d.decode()",
        "assert all(np.array_equal(p, d.plane(i)) for i, p in enumerate(info.planes(pixels)))",
        "print('SWEPT')"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    seen = dict(line.split() for line in r.stdout.splitlines() if line.startswith("hvc_"))
    assert r.returncode == 0 and "SWEPT" in r.stdout, "failed behind %s\n%s" % (list(seen)[-1:] or "the start", r.stderr[-3000:])
    assert set(seen) == {name for name, _ in calls}
    assert all(int(s) <= 0 for s in seen.values()), seen
    # what has something to refuse refuses it
    for name in ("hvc_decode_frames", "hvc_encode_frames", "hvc_jpeg_decode", "hvc_jpeg_decode_batch", "hvc_jpeg_decode_batch_gpu",
                 "hvc_jpeg_encode_batch", "hvc_huffman_encode_frames", "hvc_checksum_records"):  # (a copy of zero bytes is one)
        assert int(seen[name]) < 0, (name, seen[name])
