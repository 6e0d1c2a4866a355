"""MI355X-native JPEG block-transform path behind hardcamls/video-coding's
`jpeg/model` Decoder / Encoder API.

    csrc/            HIP kernels (gfx950) + the C ABI of include/hvc_jpeg.h
    libhvc_jpeg.so   built in-tree by csrc/Makefile (or __graft_entry__.build())
    hvc.py           ctypes binding of the C ABI (test harness / bench driver)

The product is the shared library; Python is only the harness that drives it.
There is no CPU fallback: loading fails loudly when the library is missing and
hvc.Context() raises when no gfx950 GPU is usable.
"""
from . import hvc  # noqa: F401
from .hvc import YUV_FORMATS, Component, Context, HvcError, build, lib, yuv_frame_bytes  # noqa: F401
