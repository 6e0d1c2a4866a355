"""Mutated copies of the reference's two JPEG files through the file-to-pixels entry point (hvc_jpeg_decode: host or GPU
reader, the block stage, the int64 fix-up for DCs outside int16) against the model restatement: the same planes, or both
refuse -- but for the kinds include/hvc_jpeg.h lists (a scan without any marker behind it: the model never returns; a
component of zero size: refused here).  A DC outside int16 is NOT a difference on this path: the pixels are the model's."""
import numpy as np
import pytest

from conftest import golden_bytes
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("mode,seed", [("any", 1), ("header", 2), ("scan", 3)])
def test_mutated_files_to_pixels(ctx, mode, seed):
    import video_coding_amd as m
    rng = np.random.Generator(np.random.PCG64(seed))
    base = [golden_bytes("mini.jpg"), golden_bytes("Mouse480.jpg")]
    offs = [m.hvc.jpeg_read_header(b).ecs_offset for b in base]
    agree = both_reject = no_marker = zero_size = beyond_int16 = 0
    for it in range(500):
        k = it & 1
        data = bytearray(base[k])
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(2, offs[k])) if mode == "header" else \
                int(rng.integers(offs[k], len(data) - 2)) if mode == "scan" else int(rng.integers(0, len(data)))
            kind = int(rng.integers(0, 4))
            data[pos] = [int(rng.integers(0, 256)), data[pos] ^ (1 << int(rng.integers(0, 8))), 0xFF, 0][kind]
        data = bytes(data)
        code = None
        try:
            info = m.hvc.jpeg_read_header(data)
            if info.coef_count > 1 << 22:
                continue
            info, pixels = ctx.jpeg_decode(data)
        except m.HvcError as e:
            code = e.code
        try:
            d = orc.Decoder(data)
            d.decode()
            oerr = None
        except ValueError as e:
            d, oerr = None, str(e)
        if code is not None and d is None:
            both_reject += 1
        elif code is not None:
            sizes = [(d.info(i)["decoded_width"], d.info(i)["decoded_height"]) for i in range(d.ncomp)]
            assert code == -8 and any(0 in wh for wh in sizes), (it, code, sizes)
            zero_size += 1
        elif d is None:
            assert "-12" in oerr, (it, oerr)
            no_marker += 1
        else:
            for i, plane in enumerate(info.planes(pixels)):
                assert np.array_equal(plane, d.plane(i)), (mode, it, i)
            agree += 1
            d2 = orc.Decoder(data)
            if np.abs(d2.coef_record()).max() > 32767:
                beyond_int16 += 1
    assert agree > 100 and both_reject > 50, (agree, both_reject, no_marker, zero_size, beyond_int16)
