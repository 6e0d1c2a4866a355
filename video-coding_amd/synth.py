"""Seeded synthetic inputs for the benchmarks: pixel planes whose 8x8
blocks are, half and half, smooth ramps with low noise or uniform random bytes.  Coefficient
batches are obtained by running THIS library's own forward path (k_encode) on them, so they are
valid, encoder-producible JPEG coefficients."""
import numpy as np


def synth_pixels(seed, h, w):
    """uint8 plane (h, w multiples of 8)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    bh, bw = h // 8, w // 8
    kind = rng.integers(0, 2, size=(bh, bw))
    yy, xx = np.mgrid[0:8, 0:8]
    gx = rng.integers(-12, 13, size=(bh, bw))
    gy = rng.integers(-12, 13, size=(bh, bw))
    base = rng.integers(0, 256, size=(bh, bw))
    ramp = (base[:, :, None, None] + gx[:, :, None, None] * xx + gy[:, :, None, None] * yy
            + rng.integers(-3, 4, size=(bh, bw, 8, 8)))
    noise = rng.integers(0, 256, size=(bh, bw, 8, 8))
    blk = np.where(kind[:, :, None, None] == 0, ramp, noise)
    blk = np.clip(blk, 0, 255).astype(np.uint8)
    return blk.transpose(0, 2, 1, 3).reshape(h, w)


def synth_frame_pixels(seed, planes):
    """one frame's tight pixel record for planes = [(blocks_w, blocks_h, qtab), ...]"""
    return np.concatenate([synth_pixels(seed + i, bh * 8, bw * 8).reshape(-1) for i, (bw, bh, _) in enumerate(planes)])
