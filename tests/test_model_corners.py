"""Inputs no encoder writes and the model decodes all the same -- the two corners round 3's verdict found refused by the
product, closed in round 4 (include/hvc_jpeg.h, conventions):

  * a component of ZERO width or height -- a sampling factor of zero other than the first component's, a frame
    dimension of zero: Decoder.init builds an empty plane (decoder.ml:304-345), decode_seq walks the MCUs with no block
    for it (:362-395) and only Frame.of_planes, i.e. get_yuv_frame, has no name for what comes out
    (common/src/frame.ml:42-61);
  * a Huffman table that gives a DC symbol 33 ... 62 magnitude bits: decoder.ml:81-96 reads them, and what follows is
    the model's 63-bit arithmetic, wrap-around included.

Host side only (no GPU): header geometry, coefficient records, the accept / refuse decision and get_yuv_frame's, against
the model restatement; files to pixels are tests/test_gpu_model_corners.py."""
import numpy as np
import pytest

from conftest import golden_bytes
from helpers import jpeg_optimised_tables
from oracle import orc


@pytest.fixture(scope="module")
def hvc():
    import video_coding_amd as m
    m.build()
    return m.hvc


# (h, v) per component; at least one component comes out without a block, never the first
EMPTY_PLANE_SAMPLINGS = [
    [(2, 2), (0, 1), (1, 1)],
    [(1, 1), (1, 0), (1, 1)],
    [(2, 1), (1, 1), (0, 0)],
    [(2, 2), (0, 0), (0, 0)],          # both chroma planes empty: of_planes still has no name for 16 x 16 beside 0 x 0
    [(1, 1), (0, 0)],                  # two components
    [(2, 2), (1, 1), (0, 2), (1, 1)],  # four
    [(1, 2), (0, 3), (1, 1)],
]


def empty_plane_file(sampling, w, h, seed):
    """random sparse coefficients for the components that have blocks, through tools/jpeg_opt_writer.py"""
    mh, mv = max(s[0] for s in sampling), max(s[1] for s in sampling)
    Wr, Hr = -(-w // (8 * mh)) * 8 * mh, -(-h // (8 * mv)) * 8 * mv
    nblk = sum((Wr * sh // mh // 8) * (Hr * sv // mv // 8) for sh, sv in sampling)
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    blocks[:, 0] = rng.integers(-300, 301, size=nblk)
    for b in range(nblk):
        k = rng.integers(0, 10)
        pos = rng.choice(np.arange(1, 64), size=k, replace=False)
        blocks[b, pos] = rng.integers(-60, 61, size=k)
    qt = np.stack([np.arange(1, 65), np.arange(64, 0, -1)]).astype(np.uint16)
    rec = blocks.reshape(-1)
    return jpeg_optimised_tables(w, h, sampling, qt, rec, table_sets=1), rec


def frame_decision(hvc, info, planes):
    """hvc_jpeg_get_yuv_frame on the model's own decoded planes -> the frame's bytes, or None where it refuses"""
    import video_coding_amd as m
    rec = np.concatenate([p.reshape(-1) for p in planes] + [np.zeros(0, np.uint8)])
    assert rec.size == info.pixel_bytes
    try:
        return hvc.jpeg_get_yuv_frame(info, rec)
    except m.HvcError as e:
        assert e.code == -8
        return None


@pytest.mark.parametrize("si", range(len(EMPTY_PLANE_SAMPLINGS)))
def test_a_component_without_blocks_is_walked_around(hvc, si):
    sampling = EMPTY_PLANE_SAMPLINGS[si]
    for (w, h) in ((40, 24), (97, 51)):
        jpg, rec = empty_plane_file(sampling, w, h, 31 * si + w)
        info = hvc.jpeg_read_header(jpg)
        d = orc.Decoder(jpg)
        assert info.n_comp == d.ncomp == len(sampling)
        empties = 0
        for i in range(info.n_comp):
            m, c, L = d.info(i), info.comp[i], info.layout[i]
            assert (c.decoded_width, c.decoded_height, c.actual_width, c.actual_height, c.hscale, c.vscale) == \
                   (m["decoded_width"], m["decoded_height"], m["actual_width"], m["actual_height"], m["hscale"], m["vscale"])
            assert (L.blocks_w, L.blocks_h) == (c.decoded_width // 8, c.decoded_height // 8)
            empties += L.blocks_w * L.blocks_h == 0
        assert empties >= 1
        _, got = hvc.jpeg_entropy_decode(jpg, info)
        assert np.array_equal(got, rec)
        assert np.array_equal(got, d.coef_record().astype(np.int16))
        (sa, _, ra), (sb, _, rb) = hvc.jpeg_entropy_decode2(jpg, golden_bytes("mini.jpg"))   # ... and two files in turn
        assert (sa, sb) == (0, 0) and np.array_equal(ra, rec)
        # the frame: Frame.of_planes raises for every one of these (an empty plane beside planes with samples, or too few)
        planes = [d.plane(i) for i in range(d.ncomp)]
        with pytest.raises(ValueError):
            d.get_yuv_frame()
        assert frame_decision(hvc, info, planes) is None
        crops = hvc.jpeg_get_cropped_planes(info, np.concatenate([p.reshape(-1) for p in planes]))
        assert np.array_equal(crops, np.concatenate([p.reshape(-1) for p in d.cropped_planes()]))


def test_a_zero_factor_in_the_first_component_raises(hvc):
    """decode_seq divides by components.(0)'s factors (decoder.ml:377-382): Division_by_zero; with every factor zero
    init's Int.round_up ~to_multiple_of:0 raises before that"""
    import video_coding_amd as m
    for sampling in ([(0, 1), (1, 1), (1, 1)], [(1, 0), (1, 1), (1, 1)], [(0, 0), (2, 2), (1, 1)]):
        jpg, _ = empty_plane_file(sampling, 40, 24, 7)
        with pytest.raises(ValueError):
            orc.Decoder(jpg).coef_record()
        with pytest.raises(m.HvcError) as e:
            hvc.jpeg_entropy_decode(jpg)
        assert e.value.code == -8
    base = bytearray(golden_bytes("mini.jpg"))
    sof = bytes(base).index(b"\xff\xc0")
    for k in range(3):
        base[sof + 11 + 3 * k] = 0      # every component's H and V nibble
    with pytest.raises(ValueError):
        orc.Decoder(bytes(base))
    with pytest.raises(m.HvcError) as e:
        hvc.jpeg_read_header(bytes(base))
    assert e.value.code == -8


@pytest.mark.parametrize("fn", ["mini.jpg", "Mouse480.jpg"])
@pytest.mark.parametrize("which", ["width", "height", "both"])
def test_a_frame_without_width_or_height_decodes_to_nothing(hvc, fn, which):
    """rounded_width = 0: every plane is empty, macroblocks_wide = 0, decode_seq yields no block (decoder.ml:304-395) -- and
    Frame.of_planes takes the three empty planes (0 / 2 = 0: C420, or C444 when the heights also agree)"""
    data = bytearray(golden_bytes(fn))
    sof = bytes(data).index(b"\xff\xc0")
    if which in ("height", "both"):
        data[sof + 5:sof + 7] = b"\0\0"
    if which in ("width", "both"):
        data[sof + 7:sof + 9] = b"\0\0"
    data = bytes(data)
    d = orc.Decoder(data)
    assert d.coef_record().size == 0
    info, coefs = hvc.jpeg_entropy_decode(data)
    assert info.coef_count == 0 and info.pixel_bytes == 0 and coefs.size == 0
    for i in range(3):
        m, c = d.info(i), info.comp[i]
        assert (c.decoded_width, c.decoded_height, c.actual_width, c.actual_height) == \
               (m["decoded_width"], m["decoded_height"], m["actual_width"], m["actual_height"])
    try:
        want = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
    except ValueError:
        want = None
    got = frame_decision(hvc, info, [np.zeros(0, np.uint8)])
    assert (got is None) == (want is None)
    if want is not None:
        assert got.size == want.size == 0


def test_get_yuv_frame_refuses_what_frame_of_planes_refuses(hvc):
    """Decoder.get_yuv_frame = Frame.of_planes of three crops: 4:2:0, 4:2:2 (the model's: half width, full height) and
    4:4:4 come through, every other sampling decodes (get_decoded_planes) and then has no Frame.t"""
    from test_host_entropy import UNUSUAL_SAMPLINGS, unusual_sampling_file
    for si, sampling in enumerate(UNUSUAL_SAMPLINGS + [[(2, 2), (1, 1), (1, 1)], [(2, 2), (1, 2), (1, 2)], [(1, 1)] * 3,
                                                        [(2, 1), (1, 1), (1, 1)], [(4, 2), (2, 1), (2, 1)]]):
        for (w, h) in ((40, 24), (33, 17)):
            jpg, _ = unusual_sampling_file(sampling, w, h, 900 + si)
            d = orc.Decoder(jpg)
            d.decode()
            info = hvc.jpeg_read_header(jpg)
            planes = [d.plane(i) for i in range(d.ncomp)]
            try:
                want = np.concatenate([p.reshape(-1) for p in d.get_yuv_frame()])
            except ValueError:
                want = None
            got = frame_decision(hvc, info, planes)
            assert (got is None) == (want is None), (sampling, w, h)
            if want is not None:
                assert np.array_equal(got, want)
            crops = hvc.jpeg_get_cropped_planes(info, np.concatenate([p.reshape(-1) for p in planes]))
            assert np.array_equal(crops, np.concatenate([p.reshape(-1) for p in d.cropped_planes()]))
    seen = [orc.Decoder(unusual_sampling_file(s, 40, 24, 1)[0]).chroma_subsampling()
            for s in ([(2, 2), (1, 1), (1, 1)], [(2, 2), (1, 2), (1, 2)], [(1, 1)] * 3)]
    assert seen == [420, 422, 444]


# ---------------------------------------------------------------------------------------------------------------------
# DC categories beyond 16 bits

def wide_dc_file(cats, w=16, h=16, q0=1, seed=0):
    """a 4:4:4 file whose luma DC differences have the given categories (bit lengths), one per block in scan order, with
    random magnitudes and signs; chroma DCs small.  Returns (file, the DC differences as Python ints)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nb = (w // 8) * (h // 8)
    diffs = []
    for i in range(nb):
        c = cats[i % len(cats)]
        if c == 0:
            diffs.append(0)
            continue
        mag = (1 << (c - 1)) | int(rng.integers(0, 1 << 62)) & ((1 << (c - 1)) - 1)
        diffs.append(mag if rng.integers(0, 2) else -mag)
    rec = np.zeros((3 * nb, 64), dtype=object)
    acc = 0
    for i, dv in enumerate(diffs):
        acc += dv                    # (the writer codes differences of what the record holds: plain integers here)
        rec[i, 0] = acc
        rec[i, 1 + i % 5] = int(rng.integers(-20, 21))
    rec[nb:, 0] = [int(x) for x in rng.integers(-100, 101, size=2 * nb)]
    qt = np.stack([np.full(64, q0), np.arange(1, 65)]).astype(np.uint16)
    return jpeg_optimised_tables(w, h, 444, qt, rec.reshape(-1)), diffs


def wrap63(x):
    return ((x + (1 << 62)) % (1 << 63)) - (1 << 62)


@pytest.mark.parametrize("cats", [[17, 20, 24, 31, 32], [33, 34, 40, 47], [48, 55, 61, 62], [62, 62, 62, 62], [11, 33, 0, 62]])
def test_dc_categories_up_to_62_bits_are_read_like_the_model(hvc, cats):
    """the record-returning entry points cannot hold such a DC (HVC_E_RANGE, as for every DC beyond int16); the model
    restatement reads the file, and its DCs are the 63-bit sums of the differences the file was written from"""
    import video_coding_amd as m
    jpg, diffs = wide_dc_file(cats, 32, 16, seed=sum(cats))
    d = orc.Decoder(jpg)
    rec = d.coef_record().reshape(-1, 64)
    acc, want = 0, []
    for dv in diffs:
        acc = wrap63(acc + dv)
        want.append(acc)
    assert [int(x) for x in rec[:len(diffs), 0]] == want
    with pytest.raises(m.HvcError) as e:
        hvc.jpeg_entropy_decode(jpg)
    assert e.value.code == -5
    (sa, _, _), (sb, _, rb) = hvc.jpeg_entropy_decode2(jpg, golden_bytes("mini.jpg"))
    assert (sa, sb) == (-5, 0)


@pytest.mark.parametrize("cat", [63, 64, 100, 255])
def test_dc_categories_from_63_bits_on_have_no_model_result(hvc, cat):
    """mag' shifts by cat - 1 and cat (decoder.ml:73-79): from 63 on that is Sys.int_size or more, which OCaml leaves
    unspecified -- both sides refuse, whatever the segment holds"""
    import video_coding_amd as m
    jpg, _ = wide_dc_file([40], 16, 16)
    data = bytearray(jpg)
    at = 0
    patched = 0
    while True:       # the luma DC table (class 0, id 0): its one value becomes `cat`
        at = bytes(data).find(b"\xff\xc4", at)
        if at < 0:
            break
        ln = int.from_bytes(data[at + 2:at + 4], "big")
        if data[at + 4] == 0x00:
            vals = at + 5 + 16
            for k in range(vals, at + 2 + ln):
                if data[k] == 40:
                    data[k] = cat
                    patched += 1
        at += 2 + ln
    assert patched == 1
    with pytest.raises(ValueError):
        orc.Decoder(bytes(data)).coef_record()
    with pytest.raises(m.HvcError) as e:
        hvc.jpeg_entropy_decode(bytes(data))
    assert e.value.code in (-8, -5)
