"""End-to-end entry points through the C ABI on the GPU: whole-file decode
(Decoder.decode_a_frame), whole-frame encode (Encoder.encode_4xx) and the batch
pipeline of BASELINE config 3, against the CPU oracle and the reference's golden
files / PSNR pins."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import synth_pixels
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import video_coding_amd as hvc
    c = hvc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
def test_decode_a_frame(ctx, fn):
    import video_coding_amd as hvc
    data = golden_bytes(fn)
    info, pixels = ctx.jpeg_decode(data)
    d = orc.Decoder(data)
    d.decode()
    for i, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, d.plane(i)), (fn, i)          # get_decoded_planes
    want = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
    assert np.array_equal(hvc.hvc.jpeg_get_yuv_frame(info, pixels), want)  # get_yuv_frame


@pytest.mark.parametrize("idx", range(5))
def test_g4_psnr_pins_through_the_product_path(ctx, idx):
    """jpeg/test/model-encode-and-decode.t with BOTH directions on the GPU path: encode the reference
    frame, decode it, PSNR against the source printed like `oyuv compare psnr` -- equals the pin."""
    import video_coding_amd as hvc
    c = golden_json("g4_psnr_pins.json")["cases"][idx]
    w, h, chroma = c["width"], c["height"], c["chroma"]
    y, u, v = orc.split_yuv(golden_bytes(c["file"]), w, h, chroma)
    jpg = ctx.jpeg_encode(y, u, v, w, h, chroma, c["quality"])
    assert jpg == orc.encode_yuv(y, u, v, w, h, chroma, c["quality"])
    info, pixels = ctx.jpeg_decode(jpg)
    frame = hvc.hvc.jpeg_get_yuv_frame(info, pixels)
    planes, off = [], 0
    for i in range(3):
        n = info.comp[i].actual_width * info.comp[i].actual_height
        planes.append(frame[off:off + n].reshape(info.comp[i].actual_height, info.comp[i].actual_width))
        off += n
    got = [orc.ocaml_float_to_string(orc.psnr(a, b)) for a, b in zip((y, u, v), planes)]
    assert got == c["psnr"]


def test_g3_encode_mini_jpg_byte_exact(ctx):
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    assert ctx.jpeg_encode(y, u, v, 64, 64, 420, 75) == golden_bytes("mini.jpg")


@pytest.mark.parametrize("w,h,chroma,q", [(52, 44, 420, 95), (130, 70, 422, 40), (33, 17, 444, 80), (1920, 1080, 420, 75)])
def test_encode_decode_various_sizes(ctx, w, h, chroma, q):
    cw, ch = orc.chroma_dims(chroma, w, h)
    r8 = lambda x: (x + 7) // 8 * 8
    y = synth_pixels(1 + w, r8(h), r8(w))[:h, :w]
    u = synth_pixels(2 + w, r8(ch), r8(cw))[:ch, :cw]
    v = synth_pixels(3 + w, r8(ch), r8(cw))[:ch, :cw]
    jpg = ctx.jpeg_encode(y, u, v, w, h, chroma, q)
    assert jpg == orc.encode_yuv(y, u, v, w, h, chroma, q)
    info, pixels = ctx.jpeg_decode(jpg)
    d = orc.Decoder(jpg)
    d.decode()
    for i, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, d.plane(i))


def _make_jpegs(n, w, h, q=75):
    out = []
    for f in range(n):
        y = synth_pixels(100 + f, h, w)
        u = synth_pixels(200 + f, h // 2, w // 2)
        v = synth_pixels(300 + f, h // 2, w // 2)
        out.append(orc.encode_yuv(y, u, v, w, h, 420, q))
    return out


@pytest.mark.parametrize("gpu_entropy", [False, True])
@pytest.mark.parametrize("threads,chunk", [(1, 1), (4, 3), (8, 32)])
def test_batch_pipeline_host_output(ctx, threads, chunk, gpu_entropy):
    import video_coding_amd as hvc
    jpegs = _make_jpegs(11, 96, 64)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    stride = info.pixel_bytes
    pixels = np.zeros(len(jpegs) * stride, dtype=np.uint8)
    st = ctx.jpeg_decode_batch(jpegs, pixels, stride, threads=threads, frames_per_chunk=chunk, gpu_entropy=gpu_entropy)
    assert st.chunks == (len(jpegs) + min(chunk, len(jpegs)) - 1) // min(chunk, len(jpegs))
    for f, j in enumerate(jpegs):
        d = orc.Decoder(j)
        d.decode()
        for i, plane in enumerate(info.planes(pixels[f * stride:(f + 1) * stride])):
            assert np.array_equal(plane, d.plane(i)), (f, i)


def test_batch_pipeline_device_output_and_reuse(ctx):
    import torch
    import video_coding_amd as hvc
    jpegs = _make_jpegs(9, 64, 48, q=50)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    stride = info.pixel_bytes
    for rep in range(2):  # second call reuses the pinned ring
        d_pix = torch.zeros(len(jpegs) * stride, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.jpeg_decode_batch(jpegs, d_pix, stride, threads=3, frames_per_chunk=4)
        got = d_pix.cpu().numpy()
        for f, j in enumerate(jpegs):
            d = orc.Decoder(j)
            d.decode()
            for i, plane in enumerate(info.planes(got[f * stride:(f + 1) * stride])):
                assert np.array_equal(plane, d.plane(i)), (rep, f, i)


def test_batch_rejects_mixed_geometry(ctx):
    import video_coding_amd as hvc
    jpegs = _make_jpegs(2, 64, 48) + _make_jpegs(1, 32, 32)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    pixels = np.zeros(3 * info.pixel_bytes, dtype=np.uint8)
    with pytest.raises(hvc.HvcError):
        ctx.jpeg_decode_batch(jpegs, pixels, info.pixel_bytes, threads=2, frames_per_chunk=2)


def _widen_dqt(jpg: bytes) -> bytes:
    """Rewrite every 8-bit DQT segment of a JPEG as a 16-bit one (Pq = 1) with the same values."""
    out, i = bytearray(), 0
    while i < len(jpg):
        if jpg[i] == 0xFF and i + 1 < len(jpg) and jpg[i + 1] == 0xDB:
            ln = (jpg[i + 2] << 8) | jpg[i + 3]
            assert ln == 67 and (jpg[i + 4] >> 4) == 0
            tq = jpg[i + 4] & 15
            out += b"\xff\xdb" + (3 + 128).to_bytes(2, "big") + bytes([0x10 | tq])
            for q in jpg[i + 5:i + 5 + 64]:
                out += bytes([0, q])
            i += 2 + ln
        elif jpg[i] == 0xFF and i + 1 < len(jpg) and jpg[i + 1] == 0xDA:
            out += jpg[i:]
            break
        else:
            out.append(jpg[i])
            i += 1
    return bytes(out)


def test_sixteen_bit_dqt_file(ctx):
    """Markers.Dqt.decode reads element_precision = 8 lsl Pq (markers.ml:162-167): a file with 16-bit
    tables decodes to the same planes."""
    jpg16 = _widen_dqt(golden_bytes("mini.jpg"))
    assert jpg16 != golden_bytes("mini.jpg")
    info, pixels = ctx.jpeg_decode(jpg16)
    d = orc.Decoder(jpg16)
    d.decode()
    d8 = orc.Decoder(golden_bytes("mini.jpg"))
    d8.decode()
    for i, plane in enumerate(info.planes(pixels)):
        assert np.array_equal(plane, d.plane(i)) and np.array_equal(plane, d8.plane(i))


def test_single_component_scan(ctx):
    """Decoder.init / decode_seq handle any number of scan components (decoder.ml:304-345, 374-395):
    a monochrome file (Encoder.encode_monochrome, encoder.ml:543-551) has one 1x1 component."""
    y = synth_pixels(5, 40, 72)
    jpg = orc.encode_yuv(y, y[:1, :1], y[:1, :1], 72, 40, 400, 80)
    info, pixels = ctx.jpeg_decode(jpg)
    assert info.n_comp == 1
    d = orc.Decoder(jpg)
    d.decode()
    assert np.array_equal(info.planes(pixels)[0], d.plane(0))


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
def test_decode_a_frame_to_444(ctx, fn):
    """model.exe decode frame + oyuv convert 420 -> 444 (Planar_444.of_420) in one call"""
    data = golden_bytes(fn)
    info, frame = ctx.jpeg_decode_yuv444(data)
    d = orc.Decoder(data)
    d.decode()
    y, u, v = d.get_yuv_frame()
    assert np.array_equal(frame[0], y)
    assert np.array_equal(frame[1], orc.supersample_hv2(u))
    assert np.array_equal(frame[2], orc.supersample_hv2(v))


def test_decode_to_444_rejects_other_samplings(ctx):
    import video_coding_amd as hvc
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.444"), 64, 64, 444)
    with pytest.raises(hvc.HvcError):
        ctx.jpeg_decode_yuv444(orc.encode_yuv(y, u, v, 64, 64, 444, 75))
    yc, uc, vc = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    with pytest.raises(hvc.HvcError):  # odd size: Yuv.assert_is_420 fails in the model's tool chain
        ctx.jpeg_decode_yuv444(orc.encode_yuv(yc[:44, :51], uc[:22, :25], vc[:22, :25], 51, 44, 420, 75))


@pytest.mark.parametrize("chroma,w,h,n,threads,chunk", [(420, 96, 64, 11, 4, 3), (422, 70, 50, 7, 2, 1), (444, 33, 17, 5, 8, 16),
                                                        (420, 52, 44, 9, 3, 2), (420, 1920, 1080, 5, 8, 2)])
@pytest.mark.parametrize("gpu_entropy", [False, True])
def test_encode_batch_is_byte_identical_to_the_model(ctx, chroma, w, h, n, threads, chunk, gpu_entropy):
    """hvc_jpeg_encode_batch (config 5 end to end): every file equals Encoder.encode_4xx of its frame."""
    cw, ch = orc.chroma_dims(chroma, w, h)
    r8 = lambda x: (x + 7) // 8 * 8
    frames, want = [], []
    for f in range(n):
        y = synth_pixels(400 + f, r8(h), r8(w))[:h, :w]
        u = synth_pixels(500 + f, r8(ch), r8(cw))[:ch, :cw]
        v = synth_pixels(600 + f, r8(ch), r8(cw))[:ch, :cw]
        frames.append(np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]))
        want.append(orc.encode_yuv(y, u, v, w, h, chroma, 70))
    for rep in range(2):  # second call reuses the pinned rings
        got, st = ctx.jpeg_encode_batch(frames, w, h, chroma, 70, threads=threads, frames_per_chunk=chunk,
                                        gpu_entropy=gpu_entropy)
        assert st.chunks == (n + min(chunk, n) - 1) // min(chunk, n)
        for f in range(n):
            assert got[f] == want[f], (rep, f)


def test_encode_batch_then_decode_batch_round_trip(ctx):
    import video_coding_amd as hvc
    w, h, n = 128, 80, 13
    frames = [np.concatenate([synth_pixels(700 + f, h, w).reshape(-1), synth_pixels(800 + f, h // 2, w // 2).reshape(-1),
                              synth_pixels(900 + f, h // 2, w // 2).reshape(-1)]) for f in range(n)]
    jpegs, _ = ctx.jpeg_encode_batch(frames, w, h, 420, 85, threads=4, frames_per_chunk=4)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    pixels = np.zeros(n * info.pixel_bytes, dtype=np.uint8)
    ctx.jpeg_decode_batch(jpegs, pixels, info.pixel_bytes, threads=4, frames_per_chunk=5)
    for f in range(n):
        d = orc.Decoder(jpegs[f])
        d.decode()
        for i, plane in enumerate(info.planes(pixels[f * info.pixel_bytes:(f + 1) * info.pixel_bytes])):
            assert np.array_equal(plane, d.plane(i))


def test_encode_batch_reports_a_too_small_output_buffer(ctx):
    import ctypes as C
    import video_coding_amd as hvc
    w, h = 64, 64
    frame = np.frombuffer(golden_bytes("mini64x64.420"), dtype=np.uint8)
    out = np.empty(100, dtype=np.uint8)  # mini.jpg is 1.2 kB
    fp = (C.c_void_p * 1)(frame.ctypes.data)
    op = (C.c_void_p * 1)(out.ctypes.data)
    caps = (C.c_size_t * 1)(100)
    sizes = (C.c_size_t * 1)()
    rc = hvc.lib().hvc_jpeg_encode_batch(ctx._h, fp, 1, w, h, 420, 75, 2, 1, op, caps, sizes, None)
    assert rc == -1 and sizes[0] == len(golden_bytes("mini.jpg"))


@pytest.mark.parametrize("device", [False, True])
def test_batch_pipeline_to_444(ctx, device):
    """hvc_jpeg_decode_batch_yuv444: every frame == decode_a_frame + Planar_444.of_420 of its file."""
    import torch
    jpegs = _make_jpegs(10, 112, 80, q=65)
    w, h = 112, 80
    fs = 3 * w * h
    if device:
        out = torch.zeros(len(jpegs) * fs, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
    else:
        out = np.zeros(len(jpegs) * fs, dtype=np.uint8)
    st = ctx.jpeg_decode_batch(jpegs, out, fs, threads=3, frames_per_chunk=4, yuv444=True)
    assert st.chunks == 3
    got = out.cpu().numpy() if device else out
    for f, j in enumerate(jpegs):
        d = orc.Decoder(j)
        d.decode()
        y, u, v = d.get_yuv_frame()
        want = np.concatenate([y.reshape(-1), orc.supersample_hv2(u).reshape(-1), orc.supersample_hv2(v).reshape(-1)])
        assert np.array_equal(got[f * fs:(f + 1) * fs], want), f


def test_batch_to_444_rejects_444_files(ctx):
    import video_coding_amd as hvc
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.444"), 64, 64, 444)
    j = orc.encode_yuv(y, u, v, 64, 64, 444, 75)
    with pytest.raises(hvc.HvcError):
        ctx.jpeg_decode_batch([j, j], np.zeros(2 * 3 * 64 * 64, dtype=np.uint8), 3 * 64 * 64, yuv444=True)


def test_encode_refuses_frames_the_model_cannot_walk(ctx):
    """4:2:0 at width 16k + 1: the model raises "[Plane.get] out of bounds"; the library returns an error
    before touching the GPU (single frame and batch)."""
    import video_coding_amd as hvc
    y, u, v = np.zeros((9, 17), np.uint8), np.zeros((4, 8), np.uint8), np.zeros((4, 8), np.uint8)
    with pytest.raises(hvc.HvcError):
        ctx.jpeg_encode(y, u, v, 17, 9, 420, 75)
    with pytest.raises(hvc.HvcError):
        ctx.jpeg_encode_batch([np.zeros(17 * 9 + 2 * 8 * 4, np.uint8)], 17, 9, 420, 75)


@pytest.mark.parametrize("yuv444", [False, True])
def test_batch_pipeline_with_the_huffman_reader_on_the_gpu(ctx, yuv444):
    """hvc_jpeg_decode_batch_gpu, device output, several chunks, ring reuse; 1080p-sized frames so that a
    frame spans many workgroups of subsequences; every frame against the oracle."""
    import torch
    jpegs = _make_jpegs(7, 640, 352, q=80)
    w, h = 640, 352
    import video_coding_amd as hvc
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    fs = 3 * w * h if yuv444 else info.pixel_bytes
    for rep in range(2):
        out = torch.zeros(len(jpegs) * fs, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        st = ctx.jpeg_decode_batch(jpegs, out, fs, threads=3, frames_per_chunk=2, yuv444=yuv444, gpu_entropy=True)
        assert st.entropy_ms_sum == 0  # the GPU pipeline ran (the host-decoder pipeline reports its entropy time)
        got = out.cpu().numpy()
        for f, j in enumerate(jpegs):
            d = orc.Decoder(j)
            d.decode()
            if yuv444:
                y, u, v = d.get_yuv_frame()
                want = np.concatenate([y.reshape(-1), orc.supersample_hv2(u).reshape(-1), orc.supersample_hv2(v).reshape(-1)])
                assert np.array_equal(got[f * fs:(f + 1) * fs], want), (rep, f)
            else:
                for i, plane in enumerate(info.planes(got[f * fs:(f + 1) * fs])):
                    assert np.array_equal(plane, d.plane(i)), (rep, f, i)


def test_gpu_batch_falls_back_to_the_host_pipeline_for_special_streams(ctx):
    """a truncated file inside the batch: the model reads zero bits past the end; the GPU decoder reports
    'stream ends early' and the call is redone by the host-decoder pipeline -- same planes as the oracle"""
    import video_coding_amd as hvc
    jpegs = _make_jpegs(5, 96, 64)
    info = hvc.hvc.jpeg_read_header(jpegs[2])
    jpegs[2] = jpegs[2][:info.ecs_offset + 300] + b"\xff\xd9"
    stride = info.pixel_bytes
    pixels = np.zeros(len(jpegs) * stride, dtype=np.uint8)
    st = ctx.jpeg_decode_batch(jpegs, pixels, stride, threads=2, frames_per_chunk=2, gpu_entropy=True)
    assert st.entropy_ms_sum > 0  # the host-decoder pipeline produced the result
    for f, j in enumerate(jpegs):
        d = orc.Decoder(j)
        d.decode()
        for i, plane in enumerate(info.planes(pixels[f * stride:(f + 1) * stride])):
            assert np.array_equal(plane, d.plane(i)), (f, i)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("yuv444", [False, True])
def test_gpu_batch_redoes_only_what_it_must_with_the_host_reader(ctx, device, yuv444):
    """11 files in chunks of 2; one truncated file (found out on the GPU: 'stream ends early') and one whose Huffman
    tables are no prefix code any more... no: one with a corrupted DHT length byte the host parser still accepts
    is hard to make, so the second odd one is another truncation in the last, short chunk.  The chunks holding them are
    redone by the host-reader pipeline, the others stay with the GPU reader; every frame equals the host pipeline's."""
    import torch
    import video_coding_amd as hvc
    jpegs = _make_jpegs(11, 96, 64)
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    for f in (3, 10):
        jpegs[f] = jpegs[f][:info.ecs_offset + 200 + 30 * f] + b"\xff\xd9"
    fs = 3 * info.width * info.height if yuv444 else info.pixel_bytes
    if device:
        a = torch.zeros(len(jpegs) * fs, dtype=torch.uint8, device="cuda")
        b = torch.zeros(len(jpegs) * fs, dtype=torch.uint8, device="cuda")
    else:
        a = np.zeros(len(jpegs) * fs, dtype=np.uint8)
        b = np.zeros(len(jpegs) * fs, dtype=np.uint8)
    st = ctx.jpeg_decode_batch(jpegs, a, fs, threads=3, frames_per_chunk=2, yuv444=yuv444, gpu_entropy=True)
    ctx.jpeg_decode_batch(jpegs, b, fs, threads=3, frames_per_chunk=2, yuv444=yuv444, gpu_entropy=False)
    assert st.entropy_ms_sum > 0 and st.host_prep_ms_sum > 0   # both readers had a part in it
    assert (torch.equal(a, b) if device else np.array_equal(a, b))


def _optimised(jpeg, w, h, q, table_sets=2):
    from helpers import jpeg_optimised_tables
    qt = np.stack([orc.quant_scale(orc.quant_luma(), q), orc.quant_scale(orc.quant_chroma(), q)])
    return jpeg_optimised_tables(w, h, 420, qt, orc.Decoder(jpeg).coef_record(), table_sets)


@pytest.mark.parametrize("device", [False, True])
@pytest.mark.parametrize("yuv444", [False, True])
def test_gpu_batch_with_per_file_huffman_tables_stays_on_the_gpu(ctx, device, yuv444):
    """Every file of the batch carries Huffman tables optimised for itself (libjpeg -optimize style), two files
    the model's default tables, one three table sets -- hvc_jpeg_decode_batch_gpu keeps all of them with the GPU
    reader (per-frame tables, PF mode: no chunk falls to the host reader) and every frame equals the model's."""
    import torch
    import video_coding_amd as hvc
    w, h, q = 640, 352, 80
    plain = _make_jpegs(11, w, h, q=q)
    jpegs = [_optimised(j, w, h, q, 3 if f == 7 else 2) if f not in (2, 5) else j for f, j in enumerate(plain)]
    assert len({j[:700] for j in jpegs}) >= 10
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    fs = 3 * w * h if yuv444 else info.pixel_bytes
    for chunk in (2, 4):
        out = torch.zeros(len(jpegs) * fs, dtype=torch.uint8, device="cuda") if device else np.zeros(len(jpegs) * fs, np.uint8)
        st = ctx.jpeg_decode_batch(jpegs, out, fs, threads=3, frames_per_chunk=chunk, yuv444=yuv444, gpu_entropy=True)
        assert st.entropy_ms_sum == 0, "a chunk fell to the host reader"
        got = out.cpu().numpy() if device else out
        for f, j in enumerate(jpegs):
            d = orc.Decoder(j)
            d.decode()
            if yuv444:
                y, u, v = d.get_yuv_frame()
                want = np.concatenate([y.reshape(-1), orc.supersample_hv2(u).reshape(-1), orc.supersample_hv2(v).reshape(-1)])
                assert np.array_equal(got[f * fs:(f + 1) * fs], want), (chunk, f)
            else:
                for i, plane in enumerate(info.planes(got[f * fs:(f + 1) * fs])):
                    assert np.array_equal(plane, d.plane(i)), (chunk, f, i)


def test_gpu_batch_whose_first_file_has_three_table_sets(ctx):
    """the first file sets the batch's reference tables: three different (DC, AC) pairs do not fit the two LDS
    slots, so every chunk runs with per-component tables from device memory -- still on the GPU"""
    import video_coding_amd as hvc
    w, h, q = 96, 64, 60
    jpegs = [_optimised(j, w, h, q, 3) for j in _make_jpegs(7, w, h, q=q)]
    info = hvc.hvc.jpeg_read_header(jpegs[0])
    fs = info.pixel_bytes
    out = np.zeros(len(jpegs) * fs, np.uint8)
    st = ctx.jpeg_decode_batch(jpegs, out, fs, threads=2, frames_per_chunk=3, gpu_entropy=True)
    assert st.entropy_ms_sum == 0
    for f, j in enumerate(jpegs):
        d = orc.Decoder(j)
        d.decode()
        for i, plane in enumerate(info.planes(out[f * fs:(f + 1) * fs])):
            assert np.array_equal(plane, d.plane(i)), (f, i)


@pytest.mark.parametrize("chroma,w,h", [(444, 64, 64), (420, 96, 64), (420, 640, 352)])
def test_dc_beyond_int16_decodes_like_the_model(ctx, chroma, w, h):
    """A malformed-but-decodable stream: DC differences of +-2047 pile up far beyond int16 (the model's ints are
    63-bit, decoder.ml:143).  The int16 record cannot hold those DCs, so the file-to-pixels entry points carry them
    on a side list through an int64 fix-up: hvc_jpeg_decode, hvc_jpeg_decode_yuv444 and both batch pipelines
    (device and host output, fused 4:4:4 output) must give the model's planes, in every component."""
    import torch
    import video_coding_amd as hvc
    from helpers import jpeg_optimised_tables
    q = 60
    qt = np.stack([orc.quant_scale(orc.quant_luma(), q), orc.quant_scale(orc.quant_chroma(), q)])
    info0 = hvc.hvc.jpeg_encoder_layout(w, h, chroma, q)
    rng = np.random.Generator(np.random.PCG64(w + h + chroma))
    files = []
    for variant in range(3):
        rec = np.zeros(info0.coef_count, dtype=np.int64).reshape(-1, 64)
        rec[:, 1:6] = rng.integers(-3, 4, size=(rec.shape[0], 5))
        for i in range(3):   # per component (the writer codes differences in scan order): a walk of +-2047 steps
            L = info0.layout[i]
            nb = L.blocks_w * L.blocks_h
            steps = rng.choice([-2047, 2047, 2047, 900, 0], size=nb) if variant else np.full(nb, 2047)
            if variant == 2 and i == 1:
                steps = -np.abs(steps)
            plane = rec[L.coef_offset // 64:L.coef_offset // 64 + nb]
            # differences accumulate in SCAN order; any assignment of absolute values is a valid record, the writer
            # derives the differences -- keep each step within what the baseline categories can code (|d| <= 2047
            # between blocks that follow each other in scan order is not guaranteed by a raster walk, so clamp the
            # walk itself mildly and let the writer's category reach 12-16 bits where the scan order jumps)
            plane[:, 0] = np.cumsum(steps)
        files.append(jpeg_optimised_tables(w, h, chroma, qt, rec.reshape(-1)))
    wants = []
    for j in files:
        d = orc.Decoder(j)
        assert np.abs(d.coef_record()).max() > 40000      # far beyond int16: the record-level reader refuses
        with pytest.raises(hvc.HvcError) as e:
            hvc.hvc.jpeg_entropy_decode(j)
        assert e.value.code == -5
        d2 = orc.Decoder(j)
        d2.decode()
        wants.append(d2)
    info = hvc.hvc.jpeg_read_header(files[0])
    for j, d in zip(files, wants):                        # one file at a time
        _, pixels = ctx.jpeg_decode(j)
        for i, plane in enumerate(info.planes(pixels)):
            assert np.array_equal(plane, d.plane(i)), i
        if chroma == 420:
            _, frame = ctx.jpeg_decode_yuv444(j)
            y, u, v = d.get_yuv_frame()
            assert np.array_equal(frame[0], y) and np.array_equal(frame[1], orc.supersample_hv2(u)) and \
                np.array_equal(frame[2], orc.supersample_hv2(v))
    batch = [files[i % 3] for i in range(7)] + _plain_same_geometry(ctx, w, h, chroma, q, 2)
    order = [0, 1, 2, 0, 1, 2, 0, None, None]
    for gpu in (False, True):
        for device in (False, True):
            fs = info.pixel_bytes
            out = torch.zeros(len(batch) * fs, dtype=torch.uint8, device="cuda") if device else np.zeros(len(batch) * fs, np.uint8)
            ctx.jpeg_decode_batch(batch, out, fs, threads=3, frames_per_chunk=4, gpu_entropy=gpu)
            got = out.cpu().numpy() if device else out
            for f, j in enumerate(batch):
                d = wants[order[f]] if order[f] is not None else None
                if d is None:
                    d = orc.Decoder(j)
                    d.decode()
                for i, plane in enumerate(info.planes(got[f * fs:(f + 1) * fs])):
                    assert np.array_equal(plane, d.plane(i)), (gpu, device, f, i)
        if chroma == 420:
            fs = 3 * w * h
            out = np.zeros(len(batch) * fs, np.uint8)
            ctx.jpeg_decode_batch(batch, out, fs, threads=2, frames_per_chunk=3, yuv444=True, gpu_entropy=gpu)
            for f in range(7):
                y, u, v = wants[order[f]].get_yuv_frame()
                want = np.concatenate([y.reshape(-1), orc.supersample_hv2(u).reshape(-1), orc.supersample_hv2(v).reshape(-1)])
                assert np.array_equal(out[f * fs:(f + 1) * fs], want), (gpu, f)


def _plain_same_geometry(ctx, w, h, chroma, q, n):
    """ordinary files of the same geometry and quantiser tables (the model's encoder), to sit in the same batch"""
    cw, ch = orc.chroma_dims(chroma, w, h)
    out = []
    for f in range(n):
        out.append(orc.encode_yuv(synth_pixels(900 + f, h, w), synth_pixels(910 + f, ch, cw), synth_pixels(920 + f, ch, cw), w, h, chroma, q))
    return out
