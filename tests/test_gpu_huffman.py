"""hvc_huffman_encode_frames: the encoder's back end (RLE + Huffman + byte stuffing + flush_with_1s) as
data-parallel GPU passes.  Every segment must equal the host coder's scan data (itself byte-identical
to Model.Encoder: G3, G8, cram sessions), so header + segment + EOI is the model's file."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import every_symbol_record, synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


def host_segment(hvc, info, rec):
    """scan data of the host coder's file: between the SOS header and the EOI marker"""
    jpg = hvc.hvc.jpeg_entropy_encode(info, rec)
    head = hvc.hvc.jpeg_header(info)
    assert jpg[:len(head)] == head and jpg[-2:] == b"\xff\xd9"
    return jpg[len(head):-2]


def natural_record(info, seed, w, h, chroma, quality):
    cw, ch = orc.chroma_dims(chroma, w, h)
    r8 = lambda x: (x + 7) // 8 * 8
    y = synth_pixels(seed, r8(h), r8(w))[:h, :w]
    u = synth_pixels(seed + 1, r8(ch), r8(cw))[:ch, :cw]
    v = synth_pixels(seed + 2, r8(ch), r8(cw))[:ch, :cw]
    jpg, coefs = orc.encode_yuv(y, u, v, w, h, chroma, quality, want_coefs=True)
    return np.concatenate([c.reshape(-1) for c in coefs]), jpg


@pytest.mark.parametrize("w,h,chroma,quality", [(64, 64, 420, 75), (52, 44, 420, 95), (130, 70, 422, 40), (33, 17, 444, 80),
                                                (16, 8, 420, 50), (480, 320, 420, 20), (1920, 1080, 420, 75),
                                                (200, 120, 444, 100), (96, 64, 422, 1)])
@pytest.mark.parametrize("device", [False, True])
def test_segments_equal_the_host_coder_and_the_model(ctx, w, h, chroma, quality, device):
    import torch
    import video_coding_amd as hvc
    info = hvc.hvc.jpeg_encoder_layout(w, h, chroma, quality)
    recs, files = zip(*[natural_record(info, 300 + 10 * f, w, h, chroma, quality) for f in range(3)])
    batch = np.stack(recs)
    arg = torch.from_numpy(batch).cuda() if device else batch
    segs = ctx.huffman_encode_frames(info, arg, info.coef_count, 3)
    head = hvc.hvc.jpeg_header(info)
    for f in range(3):
        assert segs[f] == host_segment(hvc, info, recs[f]), f
        assert head + segs[f] + b"\xff\xd9" == files[f], f   # the model's file, byte for byte


def test_golden_mini_jpg(ctx):
    import video_coding_amd as hvc
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    _, coefs = orc.encode_yuv(y, u, v, 64, 64, 420, 75, want_coefs=True)
    info = hvc.hvc.jpeg_encoder_layout(64, 64, 420, 75)
    rec = np.concatenate([c.reshape(-1) for c in coefs])
    seg = ctx.huffman_encode_frames(info, rec, info.coef_count, 1)[0]
    assert hvc.hvc.jpeg_header(info) + seg + b"\xff\xd9" == golden_bytes("mini.jpg")


@pytest.mark.parametrize("chroma,w,h", [(420, 64, 48), (422, 48, 16), (444, 24, 40)])
def test_constructed_patterns(ctx, chroma, w, h):
    """ZRL chains, last coefficient at 63 / 62, all-zero blocks, extreme magnitudes and DC swings,
    0xFF-rich output (stuffing), blocks whose code fits inside one 32-bit word and blocks spanning many."""
    import video_coding_amd as hvc
    info = hvc.hvc.jpeg_encoder_layout(w, h, chroma, 50)
    nblk = info.coef_count // 64
    rng = np.random.Generator(np.random.PCG64(chroma))
    frames = []
    for f in range(4):
        blocks = np.zeros((nblk, 64), dtype=np.int16)
        for b in range(nblk):
            kind = (b + f) % 9
            if kind == 1:
                blocks[b, 63] = rng.choice([-1023, -1, 1, 1023])
            elif kind == 2:
                blocks[b, 1 + int(rng.choice([15, 16, 17, 31, 32, 47, 48, 62]))] = rng.choice([-1023, 1023, 5])
            elif kind == 3:
                blocks[b, 1:] = rng.integers(-1023, 1024, size=63)
            elif kind == 4:
                blocks[b, 62], blocks[b, 1] = 7, -3
            elif kind == 5:
                blocks[b, rng.choice(np.arange(1, 64), size=5, replace=False)] = rng.integers(-40, 41, size=5)
            elif kind == 6:
                blocks[b, 17], blocks[b, 34], blocks[b, 51] = 1, -1, 2
            elif kind == 7:
                blocks[b, 1:] = -1023  # long runs of one bits in the magnitudes
            elif kind == 8:
                blocks[b, 1:8] = rng.integers(-2, 3, size=7)
        blocks[:, 0] = np.resize(np.array([0, 1023, -1024, 1000, -1, 0, 0, 512, -1024, 1023], dtype=np.int16), nblk)
        frames.append(blocks.reshape(-1))
    segs = ctx.huffman_encode_frames(info, np.stack(frames), info.coef_count, 4)
    for f in range(4):
        assert segs[f] == host_segment(hvc, info, frames[f]), f
    assert any(b"\xff\x00" in s for s in segs)


def test_values_without_a_code_and_small_buffers_are_errors(ctx):
    import video_coding_amd as hvc
    info = hvc.hvc.jpeg_encoder_layout(16, 16, 444, 75)
    rec = np.zeros(info.coef_count, dtype=np.int16)
    rec[5] = 1024  # AC size 11: no code in the default tables
    with pytest.raises(hvc.HvcError) as e:
        ctx.huffman_encode_frames(info, rec, info.coef_count, 1)
    assert e.value.code == -5
    rec[5] = 3
    with pytest.raises(hvc.HvcError) as e:
        ctx.huffman_encode_frames(info, rec, info.coef_count, 1, out_cap=4)
    assert e.value.code == -1
    assert ctx.huffman_encode_frames(info, rec, info.coef_count, 1)[0] == host_segment(hvc, info, rec)


def test_g8_code_tables_on_the_device_every_symbol(ctx):
    """The GPU coder's code tables, read back from the context's device memory, are Tables.Encoder.dc_table / ac_table as the
    reference's own test prints them (jpeg/model/test/test_tables.ml:4-395 -> g8_code_tables.json), every symbol; and a record
    that holds every symbol of both table sets comes out of k_huff_len / k_huff_emit as the host coder's bytes."""
    import torch
    import video_coding_amd as hvc
    g = golden_json("g8_code_tables.json")
    for t, name in ((0, "luma"), (1, "chroma")):
        got = hvc.hvc.huffman_code_tables(t, ctx)
        assert got["dc"] == g["dc_" + name] and got["ac"] == g["ac_" + name], name
        assert got == hvc.hvc.huffman_code_tables(t)                 # ... and the host coder's
    info = hvc.hvc.jpeg_encoder_layout(128, 88, 444, 50)
    rec = every_symbol_record(info)
    for arg in (rec.reshape(1, -1), torch.from_numpy(rec.reshape(1, -1)).cuda()):
        seg = ctx.huffman_encode_frames(info, arg, info.coef_count, 1)[0]
        assert seg == host_segment(hvc, info, rec)
    d = orc.Decoder(hvc.hvc.jpeg_header(info) + seg + b"\xff\xd9")   # the model restatement reads it back
    assert np.array_equal(d.coef_record().astype(np.int16), rec)
