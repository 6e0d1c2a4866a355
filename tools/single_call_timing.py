"""HVC_CALL_TIMING=1: the host-side stages of hvc_jpeg_decode on one 1080p file (stderr lines of the library), for files
of several qualities.   HVC_CALL_TIMING=1 python tools/single_call_timing.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_coding_amd as hvc  # noqa: E402
from video_coding_amd.synth import synth_pixels  # noqa: E402

ctx = hvc.Context(0)
w, h = 1920, 1080
y, u, v = synth_pixels(11, 1088, 1920)[:h], synth_pixels(12, 544, 960)[:540], synth_pixels(13, 544, 960)[:540]
for q in (3, 25, 75):
    jpg = ctx.jpeg_encode(y, u, v, w, h, 420, q)
    print("quality", q, len(jpg) // 1024, "kB", file=sys.stderr, flush=True)
    for _ in range(6):
        ctx.jpeg_decode(jpg)
ctx.close()
