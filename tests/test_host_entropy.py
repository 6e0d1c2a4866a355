"""Host front end / back end of libhvc_jpeg.so (csrc/hvc_entropy.cpp): header
parse + geometry, Huffman decode into coefficient records, header writer + RLE +
Huffman encode.  Host C++ only -- runs without a GPU.  Checked against the CPU
oracle and the reference's golden files."""
import numpy as np
import pytest

from conftest import golden_bytes, golden_json
from helpers import coef_planes_from_jpeg, every_symbol_record, synth_pixels
from oracle import orc


@pytest.fixture(scope="module")
def hvc():
    import video_coding_amd as m
    m.build()
    return m.hvc


def record_planes(info, coefs):
    out = []
    for i in range(info.n_comp):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h * 64
        out.append(coefs[L.coef_offset:L.coef_offset + n].reshape(L.blocks_h, L.blocks_w, 64))
    return out


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
def test_header_geometry_and_tables_match_the_model(hvc, fn):
    data = golden_bytes(fn)
    info = hvc.jpeg_read_header(data)
    d = orc.Decoder(data)
    assert (info.width, info.height, info.n_comp) == (d.width, d.height, d.ncomp)
    for i in range(d.ncomp):
        o, c = d.info(i), info.comp[i]
        assert (c.identifier, c.hscale, c.vscale) == (o["identifier"], o["hscale"], o["vscale"])
        assert (c.decoded_width, c.decoded_height, c.actual_width, c.actual_height) == (
            o["decoded_width"], o["decoded_height"], o["actual_width"], o["actual_height"])
        assert info.qtab_array()[info.layout[i].qtab].tolist() == d.array(i, "quant_table").tolist()


def test_mouse480_header_golden(hvc):
    g = golden_json("mouse480_header.json")
    info = hvc.jpeg_read_header(golden_bytes("Mouse480.jpg"))
    assert (info.width, info.height) == (g["width"], g["height"])
    for i, (ident, h, v, tq) in enumerate(g["components_id_h_v_tq"]):
        assert (info.comp[i].identifier, info.comp[i].hscale, info.comp[i].vscale) == (ident, h, v)
        want = [t for t in g["quant_tables"] if t["table_identifier"] == tq][0]["elements"]
        assert info.qtab_array()[info.layout[i].qtab].tolist() == want
    data = golden_bytes("Mouse480.jpg")
    # the first 64 bytes of the extracted segment (test_codeblock_decoder.ml) contain no stuffed 0xff here
    assert bytes.fromhex(g["entropy_first64_hex"]).count(b"\xff") == 0
    assert data[info.ecs_offset:info.ecs_offset + 64].hex() == g["entropy_first64_hex"]


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
def test_entropy_decode_equals_model_coefficients(hvc, fn):
    """coefficient records == the model's per-block coefs (zig-zag) with the DC predictor added"""
    data = golden_bytes(fn)
    info, coefs = hvc.jpeg_entropy_decode(data)
    comps, _ = coef_planes_from_jpeg(data)
    for got, c in zip(record_planes(info, coefs), comps):
        assert np.array_equal(got, c["coefs"])


def test_g2_golden_blocks_through_the_front_end(hvc):
    g = golden_json("g2_mouse480_blocks.json")
    info, coefs = hvc.jpeg_entropy_decode(golden_bytes("Mouse480.jpg"))
    planes = record_planes(info, coefs)
    comp_of = {info.comp[i].identifier: i for i in range(info.n_comp)}
    s12 = lambda v: np.where(np.array(v) >= 2048, np.array(v) - 4096, np.array(v))
    for blk in g["blocks"]:
        c = s12(blk["coefs_lo12"])
        c[0] = blk["dc_pred_after"]
        got = planes[comp_of[blk["identifier"]]][blk["y"] // 8, blk["x"] // 8]
        assert got.tolist() == c.tolist(), blk["block_number"]


@pytest.mark.parametrize("fn,chroma", [("mini64x64.420", 420), ("mini64x64.422", 422), ("mini64x64.444", 444)])
@pytest.mark.parametrize("quality", [1, 20, 75, 100])
def test_entropy_encode_is_byte_identical_to_the_model(hvc, fn, chroma, quality):
    y, u, v = orc.split_yuv(golden_bytes(fn), 64, 64, chroma)
    want, coefs = orc.encode_yuv(y, u, v, 64, 64, chroma, quality, want_coefs=True)
    info = hvc.jpeg_encoder_layout(64, 64, chroma, quality)
    rec = np.concatenate([c.reshape(-1) for c in coefs])
    assert rec.size == info.coef_count
    assert hvc.jpeg_entropy_encode(info, rec) == want


def test_g3_mini_jpg_bytes_from_the_back_end(hvc):
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    _, coefs = orc.encode_yuv(y, u, v, 64, 64, 420, 75, want_coefs=True)
    info = hvc.jpeg_encoder_layout(64, 64, 420, 75)
    assert hvc.jpeg_entropy_encode(info, np.concatenate([c.reshape(-1) for c in coefs])) == golden_bytes("mini.jpg")


def test_g8_header_bytes_and_quant_tables(hvc):
    g = golden_json("g8_header_c420_480x320_q20.json")
    info = hvc.jpeg_encoder_layout(g["width"], g["height"], 420, g["quality"])
    zero = np.zeros(info.coef_count, dtype=np.int16)
    jpg = hvc.jpeg_entropy_encode(info, zero)
    assert jpg[:len(g["hex"]) // 2].hex() == g["hex"]
    q = golden_json("g5_quant_tables.json")
    for quality, want in q["luma_scaled"].items():
        assert hvc.quant_table(0, int(quality)).tolist() == want
    for quality in (1, 37, 50, 88, 100):
        assert hvc.quant_table(1, quality).tolist() == orc.quant_scale(orc.quant_chroma(), quality).tolist()


@pytest.mark.parametrize("w,h,chroma", [(52, 44, 420), (64, 64, 422), (17, 9, 444), (8, 8, 420), (100, 30, 422)])
def test_roundtrip_odd_sizes_front_end_inverts_back_end(hvc, w, h, chroma):
    """encoder layout == the oracle's padded planes; entropy decode of the produced file gives the
    coefficients back (decoder geometry can differ from the encoder's for 4:2:2: compare per block)."""
    cw, ch = orc.chroma_dims(chroma, w, h)
    y, u, v = synth_pixels(w * 100 + h, (h + 7) // 8 * 8, (w + 7) // 8 * 8)[:h, :w], \
        synth_pixels(7, (ch + 7) // 8 * 8, (cw + 7) // 8 * 8)[:ch, :cw], \
        synth_pixels(8, (ch + 7) // 8 * 8, (cw + 7) // 8 * 8)[:ch, :cw]
    want, coefs = orc.encode_yuv(y, u, v, w, h, chroma, 60, want_coefs=True)
    info = hvc.jpeg_encoder_layout(w, h, chroma, 60)
    dims = orc.encoder_plane_dims(chroma, w, h)
    for i, (pw, ph) in enumerate(dims):
        assert (info.comp[i].decoded_width, info.comp[i].decoded_height) == (pw, ph)
    jpg = hvc.jpeg_entropy_encode(info, np.concatenate([c.reshape(-1) for c in coefs]))
    assert jpg == want
    dinfo, dcoefs = hvc.jpeg_entropy_decode(jpg)
    comps, _ = coef_planes_from_jpeg(jpg)
    for got, c in zip(record_planes(dinfo, dcoefs), comps):
        assert np.array_equal(got, c["coefs"])


def test_get_yuv_frame_crops_like_the_model(hvc):
    y, u, v = orc.split_yuv(golden_bytes("mini64x64.420"), 64, 64, 420)
    yc, uc, vc = y[:44, :52], u[:22, :26], v[:22, :26]
    jpg = orc.encode_yuv(yc, uc, vc, 52, 44, 420, 90)
    d = orc.Decoder(jpg)
    d.decode()
    info = hvc.jpeg_read_header(jpg)
    rec = np.concatenate([d.plane(i).reshape(-1) for i in range(3)])
    assert rec.size == info.pixel_bytes
    got = hvc.jpeg_get_yuv_frame(info, rec)
    want = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
    assert np.array_equal(got, want)


def test_malformed_streams_fail_like_the_model(hvc):
    import video_coding_amd as m
    data = bytearray(golden_bytes("mini.jpg"))
    with pytest.raises(m.HvcError) as e:      # progressive SOF2 marker: "unsupported marker code"
        bad = bytes(data).replace(b"\xff\xc0", b"\xff\xc2", 1)
        hvc.jpeg_read_header(bad)
    assert e.value.code == -9
    with pytest.raises(ValueError):
        orc.Decoder(bytes(data).replace(b"\xff\xc0", b"\xff\xc2", 1))
    with pytest.raises(m.HvcError) as e:      # no SOS at all
        hvc.jpeg_read_header(bytes(data[:100]))
    assert e.value.code == -8
    # truncated entropy segment: the model reads zero bits past the end; so does the front end
    info = hvc.jpeg_read_header(bytes(data))
    # (with the EOI kept: without any marker after the scan the model's extract_entropy_coded_bits
    # never terminates, so there is no behaviour to match)
    cut = bytes(data[:info.ecs_offset + 40]) + b"\xff\xd9"
    try:
        _, coefs = hvc.jpeg_entropy_decode(cut)
        comps, _ = coef_planes_from_jpeg(cut)
        for got, c in zip(record_planes(info, coefs), comps):
            assert np.array_equal(got, c["coefs"])
    except m.HvcError:
        with pytest.raises(ValueError):
            coef_planes_from_jpeg(cut)


def test_mutated_streams_never_crash_and_agree_with_the_model_when_accepted(hvc):
    """Byte-level mutations of the reference's JPEG files (random bytes, bit flips, 0xFF): front end and model (oracle)
    accept the same files and then hold the same coefficient records -- and refuse the same files, but for two kinds
    that are written down (include/hvc_jpeg.h): a scan without any marker behind it (the model's
    extract_entropy_coded_bits never returns: no behaviour to match) and a DC outside the int16 RECORD (HVC_E_RANGE from
    the record-returning entry points; the file-to-pixels ones decode it, tests/test_gpu_jpeg_api.py).  Round 3's third
    kind -- a component of zero width or height -- is decoded like the model since round 4 (tests/test_model_corners.py).
    (The same loop runs clean under ASan/UBSan, the restatement's side too: round 3 found three places where a malformed
    header took it outside its arrays.)"""
    import video_coding_amd as m
    rng = np.random.Generator(np.random.PCG64(2024))
    agree = both_reject = no_marker = dc_range = empty_planes = 0
    for it in range(1200):
        data = bytearray(golden_bytes("mini.jpg" if it % 2 == 0 else "Mouse480.jpg"))
        for _ in range(int(rng.integers(1, 5))):
            pos = int(rng.integers(0, len(data)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                data[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                data[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                data[pos] = 0xFF
        data = bytes(data)
        code = None
        try:
            info = hvc.jpeg_read_header(data)
            if info.coef_count > 1 << 24:
                continue
            _, coefs = hvc.jpeg_entropy_decode(data, info)
        except m.HvcError as e:
            code = e.code
        try:
            d = orc.Decoder(data)
            model = d.coef_record()
            oerr = None
        except ValueError as e:
            model, oerr = None, str(e)
        if code is not None and model is None:
            both_reject += 1
        elif code is not None:       # the front end refuses what the model decodes: only a DC the record cannot hold
            assert code == -5 and np.abs(model).max() > 32767, (it, code)
            dc_range += 1
        elif model is None:          # the front end decodes what the model refuses
            assert "-12" in oerr, (it, oerr)
            no_marker += 1
        else:
            assert np.array_equal(coefs, model.astype(np.int16)), it
            agree += 1
            empty_planes += any(info.layout[i].blocks_w * info.layout[i].blocks_h == 0 for i in range(info.n_comp))
    assert agree > 500 and both_reject > 200 and empty_planes > 0, (agree, both_reject, no_marker, dc_range, empty_planes)


@pytest.mark.parametrize("chroma,w,h", [(420, 32, 16), (422, 24, 8), (444, 16, 24)])
def test_entropy_round_trip_on_constructed_coefficient_patterns(hvc, chroma, w, h):
    """Coefficient records built to hit every branch of rle / write_bits (encoder.ml:127-193): runs of
    15, 16, 17, 31, 32, 47, 48 and 62 zeros (ZRL chains), a last coefficient at index 63 (no EOB) and at 62 (EOB),
    all-zero blocks, magnitudes 1 / 1023 / -1023, DC differences spanning +-2047, dense blocks.  The
    back end's file decodes to the same record through the front end AND through the model restatement."""
    info = hvc.jpeg_encoder_layout(w, h, chroma, 50)
    nblk = info.coef_count // 64
    rng = np.random.Generator(np.random.PCG64(chroma + w))
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    gaps = [15, 16, 17, 31, 32, 47, 48, 62]
    for b in range(nblk):
        kind = b % 8
        if kind == 0:
            pass                                              # all zero: DC diff + EOB only
        elif kind == 1:
            blocks[b, 63] = rng.choice([-1023, -1, 1, 1023])  # 62 zeros then the last position: three ZRLs, no EOB
        elif kind == 2:
            g = gaps[(b // 8) % len(gaps)]
            blocks[b, 1 + g] = rng.choice([-1023, 1023, 5])   # one run of g zeros after the DC
        elif kind == 3:
            blocks[b, 1:] = rng.integers(-1023, 1024, size=63)  # dense, extreme magnitudes
        elif kind == 4:
            blocks[b, 62] = 7                                 # last coefficient at 62: EOB follows
            blocks[b, 1] = -3
        elif kind == 5:
            pos = rng.choice(np.arange(1, 64), size=5, replace=False)
            blocks[b, pos] = rng.integers(-40, 41, size=5)
        elif kind == 6:
            blocks[b, 17] = 1
            blocks[b, 34] = -1                                # exactly 16 zeros between two coefficients
            blocks[b, 51] = 2
        else:
            blocks[b, 1:8] = rng.integers(-2, 3, size=7)
    # DC values: consecutive differences must stay within the 11-bit categories the default tables code
    for i in range(info.n_comp):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h
        first = L.coef_offset // 64
        blocks[first:first + n, 0] = np.resize(np.array([0, 1023, -1024, 1000, -1, 0, 0, 512, -1024, 1023], dtype=np.int16), n)
    rec = blocks.reshape(-1)
    jpg = hvc.jpeg_entropy_encode(info, rec)
    dinfo, got = hvc.jpeg_entropy_decode(jpg)
    comps, _ = coef_planes_from_jpeg(jpg)  # the model restatement's view of the same file
    for i, (mine, c) in enumerate(zip(record_planes(dinfo, got), comps)):
        assert np.array_equal(mine, c["coefs"]), i
    # per block (raster order inside each plane is the same on both sides when the geometries agree)
    if dinfo.coef_count == info.coef_count and all(
            (dinfo.layout[i].blocks_w, dinfo.layout[i].blocks_h) == (info.layout[i].blocks_w, info.layout[i].blocks_h)
            for i in range(3)):
        assert np.array_equal(got, rec)


@pytest.mark.parametrize("w,h,chroma,ok", [(17, 9, 420, False), (33, 16, 422, False), (16, 17, 420, False),
                                           (18, 9, 420, True), (17, 9, 444, True), (33, 17, 444, True), (16, 17, 422, True)])
def test_geometries_where_the_model_raises_are_refused(hvc, w, h, chroma, ok):
    """encode_seq walks the luma MCU grid; where it reaches past a chroma plane the model raises
    "[Plane.get] out of bounds" (encoder.ml:476-505, plane.ml:43-50): width or height 16k + 1 with subsampling.
    The library refuses those frames (no out-of-plane reads), and so does the oracle."""
    import ctypes as C
    import video_coding_amd as m
    info = hvc.jpeg_encoder_layout(w, h, chroma, 75)
    assert (m.lib().hvc_jpeg_encoder_check(C.byref(info)) == 0) == ok
    cw, ch = orc.chroma_dims(chroma, w, h)
    y, u, v = np.zeros((h, w), np.uint8), np.zeros((ch, cw), np.uint8), np.zeros((ch, cw), np.uint8)
    rec = np.zeros(info.coef_count, dtype=np.int16)
    if ok:
        assert hvc.jpeg_entropy_encode(info, rec)[:2] == b"\xff\xd8"
        assert orc.encode_yuv(y, u, v, w, h, chroma, 75)[:2] == b"\xff\xd8"
    else:
        with pytest.raises(m.HvcError):
            hvc.jpeg_entropy_encode(info, rec)
        with pytest.raises(RuntimeError):
            orc.encode_yuv(y, u, v, w, h, chroma, 75)


def test_dc_beyond_int16_is_refused_not_wrapped(hvc):
    """The contract edge include/hvc_jpeg.h states: the model's ints are 63-bit (decoder.ml:143), so DC differences
    that pile up beyond int16 decode there; the int16 coefficient record cannot carry them and the entropy front end
    says HVC_E_RANGE -- it never wraps.  Pinned from both sides: the model's output for the stream (a ramp of
    absolute DCs up to 131 008 in the luma plane, every block saturated white after the 17th), the refusal, and a
    stream that stops one block short of the limit, which must decode and agree."""
    from helpers import jpeg_optimised_tables
    w = h = 64
    qt = np.stack([orc.quant_scale(orc.quant_luma(), 75), orc.quant_scale(orc.quant_chroma(), 75)])
    rec = np.zeros(3 * 64 * 64, dtype=np.int64).reshape(3, 64, 64)
    rec[0, :, 0] = 2047 * (np.arange(64) + 1)        # differences of +2047: category 11, what baseline tables can code
    rec[0, :, 5] = 3
    jpg = jpeg_optimised_tables(w, h, 444, qt, rec.reshape(-1), table_sets=2)
    d = orc.Decoder(jpg)
    model = d.coef_record()                           # the model decodes it ...
    assert model.max() == 2047 * 64 and np.array_equal(model, rec.reshape(-1))
    y = d.plane(0)
    assert (y[16:] == 255).all()                      # ... block 17 on (rows 16..): DC * q far beyond the clip
    with pytest.raises(hvc.HvcError) as e:            # the library refuses, loudly and specifically
        hvc.jpeg_entropy_decode(jpg)
    assert e.value.code == -5
    ok = rec.copy()
    ok[0, 16:, 0] = 32767                             # 16 steps of +2047 = 32752, then +15 and flat: still inside int16
    jpg2 = jpeg_optimised_tables(w, h, 444, qt, ok.reshape(-1), table_sets=2)
    _, got = hvc.jpeg_entropy_decode(jpg2)
    assert np.array_equal(got, orc.Decoder(jpg2).coef_record()) and got.max() == 32767


def test_two_files_in_turn_equal_two_files_one_after_the_other():
    """hvc_jpeg_entropy_decode2 (the batch pipelines' inner routine: two walks stepped alternately so that the core has two
    dependency chains to overlap): records and per-file status as hvc_jpeg_entropy_decode gives them -- files of
    different sizes and geometries, one of them broken (the other must still come out whole), a pair of the same file."""
    import video_coding_amd as hvc
    mini, mouse = golden_bytes("mini.jpg"), golden_bytes("Mouse480.jpg")
    rng = np.random.Generator(np.random.PCG64(5))
    y, u, v = (rng.integers(0, 256, size=s, dtype=np.uint8) for s in ((120, 200), (60, 100), (60, 100)))
    noisy = orc.encode_yuv(y, u, v, 200, 120, 420, 92)
    y422 = orc.encode_yuv(y, rng.integers(0, 256, size=(120, 100), dtype=np.uint8), rng.integers(0, 256, size=(120, 100), dtype=np.uint8), 200, 120, 422, 35)
    files = [mini, mouse, noisy, y422]
    single = [hvc.hvc.jpeg_entropy_decode(j)[1] for j in files]
    for a in range(len(files)):
        for b in range(len(files)):
            (sa, _, ra), (sb, _, rb) = hvc.hvc.jpeg_entropy_decode2(files[a], files[b])
            assert (sa, sb) == (0, 0)
            assert np.array_equal(ra, single[a]) and np.array_equal(rb, single[b]), (a, b)
            assert np.array_equal(ra, orc.Decoder(files[a]).coef_record().astype(np.int16))
    # a stream that breaks half-way (an invalid code / index out of range somewhere after the cut): its own status says
    # so, exactly as the single-file entry point does, and its partner is untouched
    info = hvc.hvc.jpeg_read_header(mouse)
    broken = bytearray(mouse)
    for i in range(info.ecs_offset + 3000, info.ecs_offset + 3400):
        broken[i] = 0xFF if i % 2 == 0 else 0x00   # (0xFF 0x00 = a stuffed 0xFF: all-ones data, codes that do not exist)
    broken = bytes(broken)
    try:
        hvc.hvc.jpeg_entropy_decode(broken)
        want = 0
    except hvc.HvcError as e:
        want = e.code
    assert want != 0
    for first in (True, False):
        pair = (broken, noisy) if first else (noisy, broken)
        (sa, _, ra), (sb, _, rb) = hvc.hvc.jpeg_entropy_decode2(*pair)
        assert (sa, sb) == ((want, 0) if first else (0, want))
        assert np.array_equal(rb if first else ra, single[2])


def test_groups_of_four_symbols_end_in_every_slot(hvc):
    """The reader decodes AC symbols in groups of four behind one refill (csrc/hvc_entropy.cpp, HVC_AC_GROUP); a group ends
    early at an end of block, at the block's 64th coefficient, and at a symbol the one-lookup table does not cover (a long
    code or a large magnitude: two-step path after a refill of its own).  Blocks built so that each of these lands in each
    of the four slots, behind 0..7 short symbols, with short and long symbols alternating, and with 26-bit symbols (16-bit
    code + 10 magnitude bits) back to back; alone and in turn with a second file; against the model restatement."""
    info = hvc.jpeg_encoder_layout(64, 48, 420, 50)
    nblk = info.coef_count // 64
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    for b in range(nblk):
        kind, lead = b % 6, (b // 6) % 8
        blocks[b, 1:1 + lead] = np.where(np.arange(lead) % 2 == 0, 1, -1)  # `lead` short symbols (2-bit code + 1 bit)
        nxt = 1 + lead
        if kind == 0:
            pass                                     # ... then the end of block
        elif kind == 1:
            blocks[b, nxt] = 1023                    # ... then a large magnitude (two-step path), then the end of block
        elif kind == 2:
            blocks[b, nxt + 9] = -700                # ... then run 9 / size 10: a 16-bit code + 10 bits
            blocks[b, nxt + 10] = 2
        elif kind == 3:
            blocks[b, nxt:] = 1                      # ... short symbols up to the 64th coefficient (no end of block)
        elif kind == 4:
            blocks[b, nxt:64:2] = 1                  # ... run-1 symbols; the last coefficient may or may not be index 63
            blocks[b, nxt + 1:64:4] = -1000
        else:
            blocks[b, nxt:nxt + 12] = [-1023, 1, 1023, -1, 600, 2, -600, 1, 1, 1, 900, -900]
    for i in range(info.n_comp):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h
        blocks[L.coef_offset // 64:L.coef_offset // 64 + n, 0] = np.resize(np.array([3, -60, 1000, -1000, 0], dtype=np.int16), n)
    rec = blocks.reshape(-1)
    jpg = hvc.jpeg_entropy_encode(info, rec)
    _, got = hvc.jpeg_entropy_decode(jpg)
    assert np.array_equal(got, rec)
    assert np.array_equal(got, orc.Decoder(jpg).coef_record().astype(np.int16))
    other = golden_bytes("Mouse480.jpg")
    _, want_other = hvc.jpeg_entropy_decode(other)
    for pair in ((jpg, other), (other, jpg), (jpg, jpg)):
        (sa, _, ra), (sb, _, rb) = hvc.jpeg_entropy_decode2(*pair)
        assert (sa, sb) == (0, 0)
        assert np.array_equal(ra, rec if pair[0] is jpg else want_other)
        assert np.array_equal(rb, rec if pair[1] is jpg else want_other)
    # cut anywhere inside the last blocks: the rest reads as zero bits (bitstream_reader.ml:19-22) -- the same record as
    # the model's, whatever the reader had loaded ahead
    dinfo = hvc.jpeg_read_header(jpg)
    for cut in (1, 2, 3, 5, 8, 13, 40, 200):
        short = jpg[:len(jpg) - 2 - cut] + b"\xff\xd9"  # `cut` bytes of the segment gone, the EOI marker kept (without a
        # marker behind the scan the model's extract_entropy_coded_bits never ends: no behaviour to match)
        try:
            mine = hvc.jpeg_entropy_decode(short)[1]
        except hvc.HvcError as e:
            mine = e.code
        try:
            model = orc.Decoder(short).coef_record().astype(np.int16)
        except Exception:
            model = None
        if model is None:
            assert isinstance(mine, int), cut
        else:
            assert not isinstance(mine, int) and np.array_equal(mine, model), cut
    assert dinfo.coef_count == info.coef_count


UNUSUAL_SAMPLINGS = [
    [(1, 2), (1, 1), (1, 1)],          # 4:4:0
    [(4, 1), (1, 1), (1, 1)],          # 4:1:1
    [(1, 4), (1, 2), (1, 1)],
    [(2, 2), (2, 1), (1, 2)],          # every component its own factors
    [(1, 1), (2, 2), (2, 2)],          # the FIRST component is not the largest (the MCU grid comes from it, decoder.ml:362-373)
    [(2, 1), (1, 2), (2, 2)],
    [(3, 1), (1, 1), (1, 1)],          # a factor that is no power of two
    [(3, 2), (1, 2), (3, 1)],
    [(4, 4), (2, 2), (1, 1)],          # 21 blocks per MCU
    [(2, 2)],                          # one component with factors (its MCU is still 2 x 2 blocks, decoder.ml:374-395)
    [(2, 1), (1, 1)],                  # two components
    [(2, 2), (1, 1), (1, 1), (2, 2)],  # four components
]


def unusual_sampling_file(sampling, w, h, seed):
    """a file of that sampling with random sparse coefficients, through tools/jpeg_opt_writer.py -> (file, its record)"""
    from helpers import jpeg_optimised_tables
    mh, mv = max(s[0] for s in sampling), max(s[1] for s in sampling)
    Wr, Hr = -(-w // (8 * mh)) * 8 * mh, -(-h // (8 * mv)) * 8 * mv
    nblk = sum((Wr * sh // mh // 8) * (Hr * sv // mv // 8) for sh, sv in sampling)
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    blocks[:, 0] = rng.integers(-300, 301, size=nblk)
    for b in range(nblk):
        k = rng.integers(0, 12)
        pos = rng.choice(np.arange(1, 64), size=k, replace=False)
        blocks[b, pos] = rng.integers(-60, 61, size=k)
    qt = np.stack([np.arange(1, 65), np.arange(64, 0, -1)]).astype(np.uint16)
    rec = blocks.reshape(-1)
    return jpeg_optimised_tables(w, h, sampling, qt, rec, table_sets=min(3, len(sampling))), rec


@pytest.mark.parametrize("si", range(len(UNUSUAL_SAMPLINGS)))
def test_sampling_factors_the_encoder_never_writes(hvc, si):
    """Decoder.init / decode_seq (decoder.ml:294-345, 362-395) take any sampling factors: 4:4:0, 4:1:1, a first component
    that is not the largest, factors of three, one, two and four components.  Header geometry and coefficient record
    against the model restatement, on frame sizes that are no multiple of the MCU."""
    sampling = UNUSUAL_SAMPLINGS[si]
    for (w, h) in ((40, 24), (97, 51)):
        jpg, rec = unusual_sampling_file(sampling, w, h, 100 * si + w)
        info = hvc.jpeg_read_header(jpg)
        d = orc.Decoder(jpg)
        assert info.n_comp == d.ncomp == len(sampling)
        for i in range(info.n_comp):
            m = d.info(i)
            c = info.comp[i]
            assert (c.decoded_width, c.decoded_height, c.actual_width, c.actual_height, c.hscale, c.vscale) == \
                   (m["decoded_width"], m["decoded_height"], m["actual_width"], m["actual_height"], m["hscale"], m["vscale"])
        _, got = hvc.jpeg_entropy_decode(jpg, info)
        assert np.array_equal(got, rec)
        assert np.array_equal(got, d.coef_record().astype(np.int16))
        (sa, _, ra), (sb, _, rb) = hvc.jpeg_entropy_decode2(jpg, golden_bytes("mini.jpg"))
        assert (sa, sb) == (0, 0) and np.array_equal(ra, rec)


@pytest.mark.parametrize("chroma,w,h", [(420, 64, 48), (444, 40, 24)])
def test_coder_output_full_of_ff_bytes(hvc, chroma, w, h):
    """The coder writes its bits to a scratch buffer without stuffing and stuffs per MCU row (csrc/hvc_entropy.cpp,
    BitWriter / append_stuffed): records whose fields are mostly one-bits -- magnitudes 2^s - 1, long codes, ZRL runs -- so
    that 0xFF bytes come singly, in runs, at the 16-byte steps of the stuffing pass and at the ends of MCU rows; the file
    must decode back to the record through the front end and through the model restatement."""
    info = hvc.jpeg_encoder_layout(w, h, chroma, 50)
    nblk = info.coef_count // 64
    rng = np.random.Generator(np.random.PCG64(7 * chroma + w))
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    ones = np.array([1, 3, 7, 15, 31, 63, 127, 255, 511, 1023], dtype=np.int16)
    for b in range(nblk):
        kind = b % 5
        if kind == 0:
            blocks[b, 1:] = ones[rng.integers(0, 10, size=63)]                 # dense, every magnitude all ones
        elif kind == 1:
            blocks[b, 1:] = 1023                                                # 26-bit fields back to back
        elif kind == 2:
            pos = rng.choice(np.arange(1, 64), size=6, replace=False)
            blocks[b, pos] = ones[rng.integers(5, 10, size=6)]                  # long runs (ZRL = 11111111001) + long codes
        elif kind == 3:
            blocks[b, 63] = 1023                                                # three ZRLs, then run 14 / size 10
        else:
            blocks[b, 1:1 + (b % 40)] = -1                                      # (zeros in the magnitude bits: 0xFF from codes only)
    for i in range(info.n_comp):
        L = info.layout[i]
        n = L.blocks_w * L.blocks_h
        blocks[L.coef_offset // 64:L.coef_offset // 64 + n, 0] = np.resize(np.array([1023, -1024, 1023, 0, 2047 - 1024], dtype=np.int16), n)
    rec = blocks.reshape(-1)
    jpg = hvc.jpeg_entropy_encode(info, rec)
    scan = jpg[hvc.jpeg_read_header(jpg).ecs_offset:-2]
    assert scan.count(b"\xff\x00") > nblk            # the case is what it claims to be
    assert b"\xff\xff" not in scan                   # ... and every 0xFF is followed by its 0x00
    _, got = hvc.jpeg_entropy_decode(jpg)
    assert np.array_equal(got, rec)
    assert np.array_equal(orc.Decoder(jpg).coef_record().astype(np.int16), rec)


def test_the_block_flush_without_avx512():
    """A finished block leaves the reader's buffer with two 64-byte streaming stores where the CPU has AVX-512 and the
    record is 64-byte aligned, with eight 16-byte ones elsewhere (HVC_NO_AVX512=1 chooses that form; the choice is made
    once per process): the reader's tests again in a child process with the switch set -- and records that are only
    16-byte / only 2-byte aligned in this one."""
    import os
    import subprocess
    import sys
    import video_coding_amd as m
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("HVC_NO_AVX512") != "1":
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.join(root, "tests", "test_host_entropy.py"), "-k",
                            "entropy_decode_equals_model or groups_of_four or two_files_in_turn or sampling_factors"],
                           env=dict(os.environ, HVC_NO_AVX512="1"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
    import ctypes as C
    data = golden_bytes("Mouse480.jpg")
    info, want = m.hvc.jpeg_entropy_decode(data)
    for shift in (16, 2, 34):   # bytes off a 64-byte boundary
        raw = np.zeros(info.coef_count * 2 + 128, dtype=np.uint8)
        off = (-raw.ctypes.data) % 64 + shift
        rec = raw[off:off + info.coef_count * 2].view(np.int16)
        assert rec.ctypes.data % 64 == shift
        assert m.hvc.lib().hvc_jpeg_entropy_decode(data, len(data), C.byref(info), rec.ctypes.data) == 0
        assert np.array_equal(rec, want), shift
        assert not raw[:off].any() and not raw[off + info.coef_count * 2:].any()


def _outcome(hvc_mod, data):
    """(product, model): each a coefficient record, or None where it raises"""
    import video_coding_amd as m
    try:
        mine = hvc_mod.jpeg_entropy_decode(data)[1]
    except m.HvcError:
        mine = None
    try:
        model = orc.Decoder(data).coef_record()
    except ValueError:
        model = None
    return mine, model


def test_segments_of_a_few_bytes_raise_where_the_model_does(hvc):
    """Bitstream_reader.show raises "out of bounds" when asked for as many bits as the WHOLE segment has, or more
    (bitstream_reader.ml:31-33): a scan cut after 0..6 bytes (or ended there by a stray marker) is refused or decoded
    exactly as the model does it -- found by differential fuzzing against the restatement."""
    for fn in ("mini.jpg", "Mouse480.jpg"):
        data = golden_bytes(fn)
        off = hvc.jpeg_read_header(data).ecs_offset
        seen = set()
        for keep in range(0, 8):
            for tail in (b"\xff\xd9", b"\xff\xc4\x00\x02\xff\xd9"):
                cut = data[:off + keep] + tail
                mine, model = _outcome(hvc, cut)
                assert (mine is None) == (model is None), (fn, keep)
                if mine is not None:
                    assert np.array_equal(mine, model.astype(np.int16)), (fn, keep)
                seen.add(mine is None)
        assert seen == {True, False}, fn   # (the shortest ones raise, the longer ones read zeros past their end)
    # ... and two in turn, one of them that short
    mouse = golden_bytes("Mouse480.jpg")
    short = mouse[:hvc.jpeg_read_header(mouse).ecs_offset + 1] + b"\xff\xd9"   # (8 bits: less than the DC table's longest code)
    (sa, _, _), (sb, _, rb) = hvc.jpeg_entropy_decode2(short, mouse)
    assert sa != 0 and sb == 0 and np.array_equal(rb, hvc.jpeg_entropy_decode(mouse)[1])


def _patch_dht(data, tclass, tid, lengths=None, values=None):
    """the file with the DHT segment (class, id) rewritten: new code counts per length and / or new values"""
    b = bytearray(data)
    i = 2
    while i + 4 <= len(b) and b[i] == 0xFF:
        m, ln = b[i + 1], (b[i + 2] << 8) | b[i + 3]
        if m == 0xC4 and b[i + 4] == ((tclass << 4) | tid):
            if lengths is not None:
                b[i + 5:i + 21] = bytes(lengths)
            if values is not None:
                b[i + 21:i + 21 + len(values)] = bytes(values)
            return bytes(b)
        if m == 0xDA:
            break
        i += 2 + ln
    raise AssertionError("no such DHT segment")


def test_tables_and_headers_the_model_raises_on(hvc):
    """Headers no encoder writes, where the model raises before or while it decodes -- and where the restatement used to
    leave its arrays (differential fuzzing, round 3): a Huffman table with more codes than its lengths have room for
    (Lut.create indexes past its array, tables.ml:490-501), a DC category of 63 bits or more (mag' shifts by Sys.int_size and beyond: unspecified in OCaml), sampling
    factors of zero in every component or in the first one (divisions by zero in Decoder.init / decode_seq).  Both refuse."""
    import video_coding_amd as m
    mini = golden_bytes("mini.jpg")
    over = _patch_dht(mini, 1, 0, lengths=[0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d][:1] + [3] + [1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7c])
    big_cat = _patch_dht(mini, 0, 0, values=[200] * 12)   # every DC code names category 200

    def sampling(data, factors):
        b = bytearray(data)
        i = b.index(b"\xff\xc0")
        for c, f in enumerate(factors):
            b[i + 11 + 3 * c] = f
        return bytes(b)

    for what, data in (("over-subscribed AC table", over), ("DC category 200", big_cat),
                       ("all factors zero", sampling(mini, [0x00, 0x00, 0x00])), ("first component 0 x 2", sampling(mini, [0x02, 0x11, 0x11])),
                       ("first component 2 x 0", sampling(mini, [0x20, 0x11, 0x11]))):
        mine, model = _outcome(hvc, data)
        assert mine is None and model is None, what
    # A LATER component with a zero factor, a frame of height (or width) zero: the model decodes the other planes around
    # an empty one (Sequence.init 0 yields nothing), or nothing at all -- and so does the library since round 4
    # (tests/test_model_corners.py has the details; round 3 refused both at the header).
    data = sampling(mini, [0x22, 0x10, 0x11])
    mine, model = _outcome(hvc, data)
    assert mine is not None and np.array_equal(mine, model.astype(np.int16))
    assert hvc.jpeg_read_header(data).layout[1].blocks_h == 0
    b = bytearray(mini)
    i = b.index(b"\xff\xc0")
    b[i + 5:i + 7] = b"\x00\x00"
    mine, model = _outcome(hvc, bytes(b))
    assert mine is not None and mine.size == model.size == 0


def test_a_callers_layout_with_unaligned_component_records(hvc):
    """hvc_jpeg_entropy_decode takes the caller's hvc_jpeg_info: component records at offsets that are multiples of two
    bytes only must decode (by the copying form of the block flush), not meet an aligned streaming store."""
    import ctypes as C
    import video_coding_amd as m
    data = golden_bytes("Mouse480.jpg")
    info, want = m.hvc.jpeg_entropy_decode(data)
    for shift in (1, 4, 8, 33):
        inf = m.hvc.jpeg_read_header(data)
        at, offs = 0, []
        for i in range(inf.n_comp):
            at += shift
            inf.layout[i].coef_offset = at
            offs.append(at)
            at += inf.layout[i].blocks_w * inf.layout[i].blocks_h * 64
        rec = np.full(at + 64, 0x5A5A, dtype=np.int16)
        assert m.hvc.lib().hvc_jpeg_entropy_decode(data, len(data), C.byref(inf), rec.ctypes.data) == -1   # (records beyond coef_count)
        inf.coef_count = at
        assert m.hvc.lib().hvc_jpeg_entropy_decode(data, len(data), C.byref(inf), rec.ctypes.data) == 0
        for i in range(inf.n_comp):
            n = info.layout[i].blocks_w * info.layout[i].blocks_h * 64
            assert np.array_equal(rec[offs[i]:offs[i] + n], want[info.layout[i].coef_offset:info.layout[i].coef_offset + n]), (shift, i)
        assert (rec[:offs[0]] == 0x5A5A).all() and (rec[at:] == 0x5A5A).all()


def test_a_callers_info_is_not_trusted(hvc):
    """An hvc_jpeg_info may have been changed between hvc_jpeg_read_header and the calls that take it: component counts,
    sampling factors, block counts and offsets that would make the reader or the coder index outside its arrays or the
    caller's record, or divide by zero, are HVC_E_INVALID_ARG."""
    import ctypes as C
    import video_coding_amd as m
    data = golden_bytes("mini.jpg")
    good = m.hvc.jpeg_read_header(data)
    rec = np.zeros(good.coef_count, dtype=np.int16)
    L = m.hvc.lib()

    def broken(change):
        inf = m.hvc.jpeg_read_header(data)
        change(inf)
        return inf

    def setn(v):
        return lambda inf: setattr(inf, "n_comp", v)

    def seth(i, v):
        return lambda inf: setattr(inf.comp[i], "hscale", v)

    def setl(i, field, v):
        return lambda inf: setattr(inf.layout[i], field, v)

    cases = [setn(0), setn(5), setn(-1), seth(0, 0), seth(0, -2), seth(1, 16), lambda inf: setattr(inf.comp[2], "vscale", 0),
             setl(0, "blocks_w", 0), setl(1, "blocks_h", -3), setl(2, "coef_offset", good.coef_count),
             setl(0, "blocks_w", 1 << 19), lambda inf: setattr(inf, "coef_count", 100)]
    # (what the READER takes since round 4, because the model's decoder does: a factor of zero, a plane without blocks --
    # decoder.ml:304-395.  In a caller's info that contradicts the file they end where the model's walk would: a zero factor
    # in the first component divides by zero, a block outside its plane is "[Plane.set] out of bounds" -- HVC_E_BAD_JPEG;
    # never an index or a division gone wrong.  The CODER refuses them all: the model's encoder has no such frame.)
    the_models_raise = {3: -8, 7: -8}
    for k, change in enumerate(cases):
        inf = broken(change)
        r = L.hvc_jpeg_entropy_decode(data, len(data), C.byref(inf), rec.ctypes.data)
        if k == 6:
            assert r in (0, -8, -5), (k, r)   # (the scan read with no block for the third component: whatever the bits give)
        else:
            assert r == the_models_raise.get(k, -1), (k, r)
        out = np.zeros(1 << 16, dtype=np.uint8)
        n = C.c_size_t()
        assert L.hvc_jpeg_entropy_encode(C.byref(inf), rec.ctypes.data, out.ctypes.data, out.size, C.byref(n)) == -1, k
    # the crop of hvc_jpeg_get_yuv_frame
    pix = np.zeros(good.pixel_bytes, dtype=np.uint8)
    out = np.zeros(good.pixel_bytes, dtype=np.uint8)
    n = C.c_size_t()
    for change in (setn(7), lambda inf: setattr(inf.comp[0], "actual_width", inf.comp[0].decoded_width + 8),
                   lambda inf: setattr(inf.comp[1], "actual_height", -1), setl(2, "plane_offset", good.pixel_bytes + 1),
                   setl(0, "stride", 8), lambda inf: setattr(inf, "pixel_bytes", 64)):
        inf = broken(change)
        assert L.hvc_jpeg_get_yuv_frame(C.byref(inf), pix.ctypes.data, out.ctypes.data, out.size, C.byref(n)) == -1
    assert L.hvc_jpeg_get_yuv_frame(C.byref(good), pix.ctypes.data, out.ctypes.data, out.size, C.byref(n)) == 0


def test_g8_code_tables_of_the_host_coder_every_symbol(hvc):
    """The host coder's code tables (hvc_huffman_code_tables, host side) are Tables.Encoder.dc_table / ac_table as the reference's
    own test prints them (jpeg/model/test/test_tables.ml:4-395 -> g8_code_tables.json), every symbol -- and what the coder EMITS
    for every symbol is what those tables say: a record holding all 160 (run, size) symbols, EOB, ZRL and the twelve DC
    categories in every component goes through the back end and comes back, coefficient for coefficient, through the model
    restatement's Huffman reader (whose look-up tables are built from the file's DHT segments by create_code_table)."""
    g = golden_json("g8_code_tables.json")
    for t, name in ((0, "luma"), (1, "chroma")):
        got = hvc.huffman_code_tables(t)
        assert got["dc"] == g["dc_" + name] and got["ac"] == g["ac_" + name], name
    info = hvc.jpeg_encoder_layout(128, 88, 444, 50)
    rec = every_symbol_record(info)
    jpg = hvc.jpeg_entropy_encode(info, rec)
    d = orc.Decoder(jpg)
    assert np.array_equal(d.coef_record().astype(np.int16), rec)
    assert np.array_equal(hvc.jpeg_entropy_decode(jpg)[1], rec)
    # every symbol really is in there: the symbols of the record, counted the way Encoder.rle cuts a block
    seen_ac, seen_dc = set(), set()
    for i in range(3):
        L = info.layout[i]
        blk = rec[L.coef_offset:L.coef_offset + L.blocks_w * L.blocks_h * 64].reshape(-1, 64).astype(np.int64)
        prev = 0
        for b in blk:
            seen_dc.add((i > 0, int(abs(int(b[0]) - prev)).bit_length()))
            prev = int(b[0])
            run = 0
            for pos in range(1, 64):
                if b[pos]:
                    while run > 15:
                        seen_ac.add((i > 0, 15, 0))
                        run -= 16
                    seen_ac.add((i > 0, run, int(abs(b[pos])).bit_length()))
                    run = 0
                else:
                    run += 1
            if run:
                seen_ac.add((i > 0, 0, 0))
    for chroma in (False, True):
        assert {k for c, k in seen_dc if c == chroma} == set(range(12))
        assert {(r, s) for c, r, s in seen_ac if c == chroma} == {(r, s) for r in range(16) for s in range(1, 11)} | {(0, 0), (15, 0)}
