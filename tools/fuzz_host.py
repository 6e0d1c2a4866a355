#!/usr/bin/env python3
"""Mutated and truncated reference files through the HOST header parser and Huffman reader (no GPU needed): meant
to be run against an AddressSanitizer build of the library's host code,

    make -C video-coding_amd/csrc asan
    ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) \
        HVC_JPEG_LIB=video-coding_amd/libhvc_asan.so python tools/fuzz_host.py

Every stream either decodes or is rejected with an hvc error; a memory error aborts the process.
"""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np
import video_coding_amd as hvc
from conftest import golden_bytes
rng = np.random.Generator(np.random.PCG64(5))
base = [golden_bytes("mini.jpg"), golden_bytes("Mouse480.jpg")]
ok = err = 0
for it in range(6000):
    b = bytearray(base[it & 1])
    mode = it % 5
    if mode == 4:   # truncation
        b = b[:int(rng.integers(2, len(b)))]
    else:
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
    b = bytes(b)
    try:
        info = hvc.hvc.jpeg_read_header(b)
        if info.coef_count > 1 << 24: continue
        hvc.hvc.jpeg_entropy_decode(b, info)
        ok += 1
    except hvc.HvcError:
        err += 1
print("decoded", ok, "rejected", err)
