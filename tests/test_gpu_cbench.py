"""The path driven from plain C (tools/cbench/hvc_cbench.c: include/hvc_jpeg.h + libhvc_jpeg.so only,
no Python / PyTorch in the process): compiled with gcc here, run on the GPU, and its result checked
against the oracle on the same LCG input."""
import json
import os
import subprocess
import zlib

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lcg_pixels(n):
    out = np.empty(n, dtype=np.uint8)
    x = 12345
    i = np.arange(n, dtype=np.uint64)
    vals = np.empty(n, dtype=np.uint32)
    for k in range(n):  # sizes here are small (a few 100 kB)
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        vals[k] = x
    out[:] = (((i >> np.uint64(3)) & np.uint64(0x7F)) + ((vals >> np.uint32(24)) & np.uint32(0x3F))).astype(np.uint8)
    return out


def test_c_program_through_the_abi(tmp_path):
    import video_coding_amd as hvc
    so = hvc.build()
    exe = tmp_path / "hvc_cbench"
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "cbench", "hvc_cbench.c"), "-o", str(exe), so,
                           "-Wl,-rpath," + os.path.dirname(so)])
    w, h, frames = 208, 120, 3
    out = subprocess.run([str(exe), str(frames), "2", str(w), str(h)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [json.loads(ln) for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    res, seam = lines[0], lines[1]
    info = hvc.hvc.jpeg_encoder_layout(w, h, 420, 75)
    pix = lcg_pixels(info.pixel_bytes * frames)[:info.pixel_bytes]
    rec = np.zeros(info.pixel_bytes, dtype=np.uint8)
    for i in range(3):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h * 64
        plane = pix[L.plane_offset:L.plane_offset + n].reshape(L.blocks_h * 8, L.blocks_w * 8)
        q = info.qtab_array()[L.qtab]
        coefs = orc.fdct_quant(plane, q, L.blocks_w, L.blocks_h)
        rec[L.plane_offset:L.plane_offset + n] = orc.dequant_idct_recon(coefs, q, L.blocks_w, L.blocks_h)
    assert res["crc32_frame0"] == zlib.crc32(rec.tobytes())
    assert res["Mpixel_s"] > 0
    # the same frames through hvc_host_alloc / hvc_decode_frames_submit / hvc_wait, all slots in flight: the same pixels
    assert seam["async_crc32_frame0"] == res["crc32_frame0"] and seam["slots"] == hvc.hvc.HVC_SLOTS
    assert seam["async_batches"] >= seam["slots"] + 1 and seam["async_Mpixel_s"] > 0
