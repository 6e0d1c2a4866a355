"""What costs the host reader its time: the same number of symbols per 1080p frame in files of different predictability --
every block alike (29 short symbols: the branch predictor learns where a block ends), block lengths random, values random
too (other code lengths), a share of long symbols (two-step path) -- one file at a time and two in turn, ns per symbol.
Host only (no GPU).   python tools/bench_reader_content.py"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import video_coding_amd as pkg  # noqa: E402

H = pkg.hvc
L = H.lib()
info = H.jpeg_encoder_layout(1920, 1080, 420, 75)
nblk = info.coef_count // 64
rng = np.random.Generator(np.random.PCG64(1))


def make(kind):
    b = np.zeros((nblk, 64), np.int16)
    if kind == "DC + EOB only":
        b[:, 0] = rng.integers(-20, 21, size=nblk)
        return b.reshape(-1)
    if kind == "3 coefficients a block":
        b[:, 0] = rng.integers(-20, 21, size=nblk)
        b[:, 1:4] = rng.integers(-3, 4, size=(nblk, 3))
        return b.reshape(-1)
    if kind == "every block alike":
        b[:, 1:30] = np.where(np.arange(29) % 2 == 0, 1, -1)
        return b.reshape(-1)
    n = rng.integers(15, 44, size=nblk)
    mask = np.arange(63)[None, :] < n[:, None]
    if kind == "lengths random":
        b[:, 1:][mask] = 1
    elif kind == "lengths + values random":
        v = rng.integers(1, 8, size=(nblk, 63)) * rng.choice([-1, 1], size=(nblk, 63))
        b[:, 1:][mask] = v[mask]
    elif kind == "+ 3 % long symbols":
        v = rng.integers(1, 8, size=(nblk, 63)) * rng.choice([-1, 1], size=(nblk, 63))
        long_ = rng.random((nblk, 63)) < 0.03
        v[long_] = rng.integers(300, 1000, size=int(long_.sum()))
        b[:, 1:][mask] = v[mask]
    return b.reshape(-1)


def best(f, n=12):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t)


for kind in ("DC + EOB only", "3 coefficients a block", "every block alike", "lengths random", "lengths + values random", "+ 3 % long symbols"):
    rec = make(kind)
    jpg = H.jpeg_entropy_encode(info, rec)
    i2 = H.jpeg_read_header(jpg)
    out, out2 = np.zeros(i2.coef_count, np.int16), np.zeros(i2.coef_count, np.int16)
    sa, sb = C.c_int(), C.c_int()

    def one():
        L.hvc_jpeg_entropy_decode(jpg, len(jpg), C.byref(i2), out.ctypes.data)

    def two():
        L.hvc_jpeg_entropy_decode2(jpg, len(jpg), C.byref(i2), out.ctypes.data, C.byref(sa), jpg, len(jpg), C.byref(i2), out2.ctypes.data, C.byref(sb))

    one()
    assert np.array_equal(out, rec)
    nsym = int((rec != 0).sum()) + nblk * 2
    t1, t2 = best(one), best(two)
    print("%-26s %7d bytes %8d symbols   alone %.2f ns/symbol %5.1f ns/block   two in turn %.2f ns/symbol %5.1f ns/block" %
          (kind, len(jpg), nsym, t1 / nsym * 1e9, t1 / nblk * 1e9, t2 / (2 * nsym) * 1e9, t2 / (2 * nblk) * 1e9), flush=True)
