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
