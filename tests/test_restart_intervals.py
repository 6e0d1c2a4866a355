"""Restart intervals (DRI + RSTn, ITU-T T.81 B.2.4.4 / E.2.4): an EXTENSION of the host reader that a caller has to ask for
(SURVEY.md 8f next-1 names it).  The model parses DRI and never looks at it again, and cuts the entropy-coded segment at
the first RSTn like at any marker (decoder.ml:56-59, 261-281) -- which stays the default of every entry point (parity);
with the extension on, a file with restart intervals decodes to the coefficient record of the same frame written without
them, and libjpeg-turbo's own DRI files decode to what libjpeg-turbo makes of them within the reference's tolerance (G9).
Host side only; files to pixels: tests/test_gpu_restart_intervals.py."""
import io

import numpy as np
import pytest

from conftest import golden_bytes
from helpers import jpeg_optimised_tables, synth_pixels
from oracle import orc


@pytest.fixture(scope="module")
def hvc():
    import video_coding_amd as m
    m.build()
    return m.hvc


def random_record(sampling, w, h, seed):
    mh, mv = max(s[0] for s in sampling), max(s[1] for s in sampling)
    Wr, Hr = -(-w // (8 * mh)) * 8 * mh, -(-h // (8 * mv)) * 8 * mv
    nblk = sum((Wr * sh // mh // 8) * (Hr * sv // mv // 8) for sh, sv in sampling)
    rng = np.random.Generator(np.random.PCG64(seed))
    blocks = np.zeros((nblk, 64), dtype=np.int16)
    blocks[:, 0] = rng.integers(-900, 901, size=nblk)
    for b in range(nblk):
        k = rng.integers(0, 14)
        blocks[b, rng.choice(np.arange(1, 64), size=k, replace=False)] = rng.integers(-200, 201, size=k)
    return blocks.reshape(-1), (Wr // (8 * mh)) * (Hr // (8 * mv))


QT = np.stack([np.arange(1, 65), np.arange(64, 0, -1)]).astype(np.uint16)


@pytest.mark.parametrize("sampling", [[(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 1)], [(4, 1), (1, 2), (2, 2)]])
def test_a_file_with_restart_intervals_decodes_to_the_record_it_was_written_from(hvc, sampling):
    for (w, h) in ((64, 48), (200, 72)):
        rec, n_mcu = random_record(sampling, w, h, w + len(sampling))
        plain = jpeg_optimised_tables(w, h, sampling, QT, rec, table_sets=min(2, len(sampling)))
        assert np.array_equal(hvc.jpeg_entropy_decode(plain)[1], rec)
        for ri in (1, 2, 3, 7, n_mcu - 1, n_mcu, n_mcu + 5):
            if ri < 1:
                continue
            f = jpeg_optimised_tables(w, h, sampling, QT, rec, table_sets=min(2, len(sampling)), restart_interval=ri)
            assert f.count(b"\xff\xdd") == 1
            got = hvc.jpeg_entropy_decode(f, restart_markers=True)[1]
            assert np.array_equal(got, rec), (sampling, w, h, ri)
            assert np.array_equal(hvc.jpeg_entropy_decode(plain, restart_markers=True)[1], rec)   # a file without DRI: the same either way


def test_the_default_is_the_models_behaviour(hvc):
    """without the extension a DRI file is read as the model reads it: the segment ends at the first RSTn, zeros from there on
    -- the same record as the model restatement's, or the same refusal"""
    import video_coding_amd as m
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], 96, 64, 5)
    for ri in (1, 4, n_mcu):
        f = jpeg_optimised_tables(96, 64, 420, QT, rec, restart_interval=ri)
        code = 0
        try:
            mine = hvc.jpeg_entropy_decode(f)[1]
        except m.HvcError as e:
            mine, code = None, e.code
        try:
            model = orc.Decoder(f).coef_record()
        except ValueError:
            model = None
        if code == -5:   # (zeros read as DC differences pile up: a DC the int16 RECORD cannot hold, include/hvc_jpeg.h)
            assert model is not None and np.abs(model).max() > 32767
            continue
        assert (mine is None) == (model is None)
        if mine is not None:
            assert np.array_equal(mine, model.astype(np.int16))
            assert ri >= n_mcu or not np.array_equal(mine, rec)     # ... which is NOT the frame, unless no marker was written


def test_libjpeg_turbo_files_with_restart_markers(hvc):
    """files written by libjpeg-turbo (behind PIL) with restart markers: the reader's coefficient records through the model
    restatement's block stage against libjpeg-turbo's own decode, within 1 per sample on the luma plane (the reference's
    tolerance against ffmpeg, jpeg/test/mouse-decode.t:10-13)"""
    Image = pytest.importorskip("PIL.Image")
    rgb = np.stack([synth_pixels(70 + i, 136, 200) for i in range(3)], axis=-1)
    for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=5), dict(restart_marker_rows=1), dict(restart_marker_rows=2)):
        for subsampling in (0, 2):
            b = io.BytesIO()
            Image.fromarray(rgb).save(b, "JPEG", quality=85, subsampling=subsampling, **kw)
            jpg = b.getvalue()
            assert jpg.count(b"\xff\xdd") == 1 and sum(jpg.count(bytes([0xff, 0xd0 + k])) for k in range(8)) > 3
            info, rec = hvc.jpeg_entropy_decode(jpg, restart_markers=True)
            L = info.layout[0]
            luma = orc.dequant_idct_recon(rec[L.coef_offset:L.coef_offset + L.blocks_w * L.blocks_h * 64], info.qtab_array()[L.qtab],
                                          L.blocks_w, L.blocks_h).reshape(L.blocks_h * 8, L.blocks_w * 8)[:136, :200]
            im = Image.open(io.BytesIO(jpg))
            im.draft("YCbCr", im.size)
            im.load()
            want = np.asarray(im)[..., 0].astype(np.int64)
            assert np.abs(luma.astype(np.int64) - want).max() <= 1, (kw, subsampling)


def test_streams_that_break_their_promise(hvc):
    """fewer markers than the DRI promises (the file cut inside an interval, a marker removed): zeros from the end of the
    data on, like every truncated scan; never a crash, never an index out of its arrays"""
    import video_coding_amd as m
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], 96, 64, 9)
    f = jpeg_optimised_tables(96, 64, 420, QT, rec, restart_interval=2)
    at = hvc.jpeg_read_header(f).ecs_offset
    rng = np.random.Generator(np.random.PCG64(1))
    for trial in range(200):
        data = bytearray(f)
        kind = trial % 4
        if kind == 0:
            data = data[:int(rng.integers(at, len(f)))] + b"\xff\xd9"
        elif kind == 1:
            pos = bytes(data).find(b"\xff\xd3", at)
            data[pos:pos + 2] = b""
        elif kind == 2:
            for _ in range(3):
                data[int(rng.integers(at, len(f) - 2))] = int(rng.integers(0, 256))
        else:
            data[int(rng.integers(at, len(f) - 2))] = 0xFF
        try:
            hvc.jpeg_entropy_decode(bytes(data), restart_markers=True)
        except m.HvcError as e:
            assert e.code in (-8, -5)


def test_a_single_interval_scan_is_cut_like_the_plain_segment(hvc):
    """with the extension on, a scan of ONE interval (MCUs <= DRI) is the plain segment for both readers -- 0xFF 0xFF
    ends it and a trailing lone 0xFF is data, exactly as without the extension (decoder.ml:261-281) -- so which reader takes
    the file never changes the records."""
    rec, n_mcu = random_record([(2, 2), (1, 1), (1, 1)], 96, 64, 21)
    f = bytearray(jpeg_optimised_tables(96, 64, 420, QT, rec, restart_interval=n_mcu))
    at = hvc.jpeg_read_header(bytes(f)).ecs_offset
    eoi = bytes(f).rindex(b"\xff\xd9")
    body = f[at:eoi]
    for mutate in ("ffff_inside", "trailing_ff", "ff_then_eoi_removed"):
        if mutate == "ffff_inside":
            k = next(i for i in range(len(body) // 2, len(body) - 1) if body[i] != 0xFF and body[i + 1] != 0xFF and body[i - 1] != 0xFF)
            g = bytes(f[:at] + body[:k] + b"\xff\xff" + body[k:] + f[eoi:])
        elif mutate == "trailing_ff":
            g = bytes(f[:at] + body + b"\xff")
        else:
            g = bytes(f[:at] + body + b"\xff\xff")
        a = hvc.jpeg_entropy_decode(g)[1]
        b = hvc.jpeg_entropy_decode(g, restart_markers=True)[1]
        assert np.array_equal(a, b), mutate
