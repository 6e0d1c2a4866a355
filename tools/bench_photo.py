#!/usr/bin/env python3
"""Photograph-like 1080p content through the GPU reader ("record rounds-to-settle and Gpixel/s"): the
reference's Mouse480.jpg enlarged to 1920 x 1080 (smooth) and the same with sensor-like noise, encoded by THIS library's
encoder at three qualities; per content and quality 256 copies through hvc_jpeg_decode_batch_gpu (device output) and one
file through hvc_jpeg_entropy_decode_gpu.  An instrumented build (-DHVC_HD_STATS, build/variants/libhvc_hdstats.so via
HVC_JPEG_LIB) additionally prints the work lists' lengths per round to stderr.  One JSON line per case; nothing here
touches oracle/ (the frames come from the library's own decoder and encoder)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import video_coding_amd as hvc  # noqa: E402


def main():
    import torch
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ctx = hvc.Context(0)
    mouse = open(os.path.join(root, "tests", "golden", "Mouse480.jpg"), "rb").read()
    info, pix = ctx.jpeg_decode(mouse)
    planes = info.planes(pix)
    y, u, v = planes[0][:320, :480], planes[1][:160, :240], planes[2][:160, :240]
    W, H = 1920, 1080
    big = [np.asarray(Image.fromarray(np.ascontiguousarray(p)).resize(s, Image.BICUBIC)) for p, s in ((y, (W, H)), (u, (W // 2, H // 2)), (v, (W // 2, H // 2)))]
    rng = np.random.Generator(np.random.PCG64(42))
    noisy = [np.clip(p.astype(np.int32) + np.rint(rng.normal(0.0, s, p.shape)).astype(np.int32), 0, 255).astype(np.uint8)
             for p, s in zip(big, (4.0, 2.0, 2.0))]
    n = 256
    for name, (yy, uu, vv) in (("smooth", big), ("noisy", noisy)):
        for q in (50, 75, 90):
            jpg = ctx.jpeg_encode(yy, uu, vv, W, H, 420, q)
            files = [jpg] * n
            fs = hvc.hvc.jpeg_read_header(jpg).pixel_bytes
            out = torch.zeros((n, fs), dtype=torch.uint8, device="cuda")
            ctx.jpeg_decode_batch(files, out, fs, threads=16, frames_per_chunk=64, gpu_entropy=True)  # warm-up: rings
            torch.cuda.synchronize()
            best, st = None, None
            for _ in range(3):
                t0 = time.perf_counter()
                s = ctx.jpeg_decode_batch(files, out, fs, threads=16, frames_per_chunk=64, gpu_entropy=True)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if best is None or dt < best:
                    best, st = dt, s
            _, _, used = ctx.jpeg_entropy_decode_gpu([jpg], device=True)
            ref_info, ref = ctx.jpeg_decode(jpg)
            same = bool(np.array_equal(out[7].cpu().numpy(), ref))
            print(json.dumps({"content": name, "quality": q, "file_kB": round(len(jpg) / 1024, 1), "files": n,
                              "bits_per_pixel": round(8 * len(jpg) / (W * H), 3),
                              "Gpixel_s": round(n * W * H / best / 1e9, 2), "call_ms": round(best * 1e3, 2),
                              "host_reader_ms_on_handed_back_chunks": round(st.entropy_ms_sum, 2),
                              "single_file_used_gpu_reader": int(used), "batch_equals_single_file_decode": same}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
