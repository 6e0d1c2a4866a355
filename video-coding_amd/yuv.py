"""Raw planar .yuv frames and the comparison metrics of the reference's cram tests.

    Frame.create / input / output      common/src/frame.ml:31-76   (Y, then U, then V, row-major)
    Chroma_subsampling.width / height  common/src/frame.ml:3-22    (integer halves)
    Ocompare.*                         tools/src/ocompare.ml:6-59

The integer sums run in libhvc_jpeg.so (hvc_compare_planes); the float metrics are the model's
expressions evaluated in the same order in IEEE doubles, and `float_to_string` prints them the way
`print_s [%sexp (x : float)]` does, so the cram files' expected lines can be compared as text.
"""
import math

import numpy as np

from . import hvc


def chroma_dims(chroma, width, height):
    """common/src/frame.ml:9-22"""
    if chroma == 420:
        return width // 2, height // 2
    if chroma == 422:
        return width // 2, height
    if chroma == 444:
        return width, height
    raise ValueError("Invalid chroma type")  # jpeg/bin/model.ml:83


def frame_bytes(chroma, width, height):
    cw, ch = chroma_dims(chroma, width, height)
    return width * height + 2 * cw * ch


def split_frame(buf, width, height, chroma=420):
    """Frame.input (frame.ml:72-76): the three planes of one raw frame as 2-D uint8 views."""
    buf = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    cw, ch = chroma_dims(chroma, width, height)
    if buf.size < width * height + 2 * cw * ch:
        raise EOFError("End_of_image")  # Plane.input raises End_of_image on a short read (plane.ml)
    y = buf[:width * height].reshape(height, width)
    u = buf[width * height:width * height + cw * ch].reshape(ch, cw)
    v = buf[width * height + cw * ch:width * height + 2 * cw * ch].reshape(ch, cw)
    return y, u, v


def read_frame(path, width, height, chroma=420, index=0):
    n = frame_bytes(chroma, width, height)
    with open(path, "rb") as f:
        f.seek(index * n)
        return split_frame(f.read(n), width, height, chroma)


def write_frame(path, planes):
    """Frame.output (frame.ml:66-70)"""
    with open(path, "wb") as f:
        for p in planes:
            f.write(np.ascontiguousarray(p, dtype=np.uint8).tobytes())


def max_difference(a, b):
    return hvc.compare_planes(a, b)[0]


def mean_difference(a, b):
    a = np.asarray(a)
    return float(hvc.compare_planes(a, b)[1]) / (float(a.shape[1]) * float(a.shape[0]))


def mean_square_error(a, b):
    a = np.asarray(a)
    return float(hvc.compare_planes(a, b)[2]) / (float(a.shape[1]) * float(a.shape[0]))


def psnr(a, b, r=255.0):
    """Float.(10. * log10 (r * r / mean_square_error f1 f2))  (ocompare.ml:57-59)"""
    mse = mean_square_error(a, b)
    q = r * r / mse if mse != 0.0 else math.inf
    return 10.0 * (math.log10(q) if q != math.inf else math.inf)


def float_to_string(x):
    """Sexp of a float (Base Float.to_string): the shortest of %.15g / %.17g that round-trips, a
    trailing '.' for integral values, INF / -INF / NAN."""
    if math.isnan(x):
        return "NAN"
    if math.isinf(x):
        return "INF" if x > 0 else "-INF"
    s = "%.15g" % x
    if float(s) != x:
        s = "%.17g" % x
    if all(c in "-0123456789" for c in s):
        s += "."
    return s
