#!/usr/bin/env python3
"""Extract the reference's own golden vectors into small data fixtures.

Run once in the development container (where /root/reference exists):

    python tests/golden/make_golden.py

It reads the reference's *test expectations and test data* (never its source
code) and writes data-only fixtures next to this script.  The fixtures are the
inputs and expected outputs the reference's own tests hold for the JPEG block
path.  Nothing here runs the reference: it is
OCaml and there is no OCaml toolchain in this image.
"""
import json
import os
import re
import shutil
import sys

REF = os.environ.get("HVC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def ints(s):
    return [int(x) for x in re.findall(r"-?\d+", s)]


def g1_chen():
    """jpeg/model/test/test_chen_dct.ml:47-87: input / fdct(/4 rounded) / idct."""
    s = read("jpeg/model/test/test_chen_dct.ml")
    m = re.search(r"input\n(.*?)fdct\n(.*?)idct\n(.*?)\|\}\]", s[s.index("[%expect"):], re.S)
    inp, fd, idc = (ints(m.group(i)) for i in (1, 2, 3))
    assert len(inp) == len(fd) == len(idc) == 64
    return {"source": "jpeg/model/test/test_chen_dct.ml:47-87", "input": inp,
            "fdct_div4_rounded": fd, "idct_of_fdct": idc}


def hexblock(s, digits):
    vals = re.findall(r"\b[0-9a-f]{%d}\b" % digits, s)
    assert len(vals) == 64, (len(vals), s)
    return [int(v, 16) for v in vals]


def g2_mouse_blocks():
    """jpeg/hardcaml/test/test_decoder_accelerator.ml:209-376: the model's
    Component.Summary for blocks 0..5 of Mouse480.jpg.  coefs/dequant are the
    low 12 bits (3 hex digits), idct/recon the low 8 bits (2 hex digits), per
    jpeg/model/src/util.ml:3-26."""
    s = read("jpeg/hardcaml/test/test_decoder_accelerator.ml")
    s = s[s.index("((width 480) (height 320))"):]
    blocks = []
    for m in re.finditer(
            r"\(\(block_number (\d+)\).*?\(comp\s*\(\(x (\d+)\) \(y (\d+)\) \(dc_pred (-?\d+)\) "
            r"\(component\.identifier (\d+)\)\s*\(coefs\s*(\(.*?\))\)\s*\(dequant\s*(\(.*?\))\)\s*"
            r"\(idct\s*(\(.*?\))\)\s*\(recon\s*(\(.*?\))\)\)\)\)", s, re.S):
        blocks.append({
            "block_number": int(m.group(1)), "x": int(m.group(2)), "y": int(m.group(3)),
            "dc_pred_after": int(m.group(4)), "identifier": int(m.group(5)),
            "coefs_lo12": hexblock(m.group(6), 3), "dequant_lo12": hexblock(m.group(7), 3),
            "idct_lo8": hexblock(m.group(8), 2), "recon": hexblock(m.group(9), 2)})
    assert [b["block_number"] for b in blocks] == [0, 1, 2, 3, 4, 5], blocks
    return {"source": "jpeg/hardcaml/test/test_decoder_accelerator.ml:209-376", "blocks": blocks}


def g4_psnr():
    """PSNR lines of jpeg/test/*.t (17 significant digits pin the integer SSE)."""
    s = read("jpeg/test/model-encode-and-decode.t")
    cases = []
    for m in re.finditer(
            r"model encode frame \.\./test_data/(\S+) (\d+)x(\d+) model\.jpg -quality (\d+)(?: -chroma (\d+))?"
            r".*?compare psnr[^\n]*\n\s+(\S+)\n\s+(\S+)\n\s+(\S+)\n", s, re.S):
        cases.append({"file": m.group(1), "width": int(m.group(2)), "height": int(m.group(3)),
                      "quality": int(m.group(4)), "chroma": int(m.group(5) or 420),
                      "psnr": [m.group(6), m.group(7), m.group(8)]})
    assert len(cases) == 5, cases
    s2 = read("jpeg/test/test-nonstandard-sizes.t")
    m = re.search(r"compare psnr[^\n]*\n\s+(\S+)\n\s+(\S+)\n\s+(\S+)\n", s2)
    nonstd = {"file": "mini64x64.420", "src_width": 64, "src_height": 64, "width": 52, "height": 44,
              "quality": 95, "chroma": 420, "psnr": [m.group(1), m.group(2), m.group(3)],
              "source": "jpeg/test/test-nonstandard-sizes.t:3-15"}
    return {"source": "jpeg/test/model-encode-and-decode.t:7-72", "cases": cases, "nonstandard": nonstd,
            "max_difference_vs_ffmpeg": {"source": "jpeg/test/mouse-decode.t:10-13", "mouse480": [1, 0, 0]}}


def g5_quant():
    s = read("jpeg/model/test/test_quant_tables.ml")
    out = {}
    for m in re.finditer(r'\("Quant\.scale Quant\.luma (\d+)"\s*\((.*?)\)\)', s, re.S):
        v = ints(m.group(2))
        assert len(v) == 64
        out[m.group(1)] = v
    assert sorted(out, key=int) == ["1", "25", "50", "75", "95", "100"], list(out)
    m = re.search(r"\(\(izz\s*\((.*?)\)\)\s*\(fzz\s*\((.*?)\)\)\)", s, re.S)
    return {"source": "jpeg/model/test/test_quant_tables.ml:4-81", "luma_scaled": out,
            "izz": ints(m.group(1)), "fzz": ints(m.group(2))}


def dump_planes(s):
    return [ints(line) for line in s.strip().splitlines()]


def g7_upsample():
    """tools/src/planar_444.ml:139-249 expect blocks (4x4 frame)."""
    s = read("tools/src/planar_444.ml")
    out = {}
    for name in ("444<->422", "444<->420"):
        t = s[s.index('let%%expect_test "%s"' % name):]
        dumps = re.findall(r"\{\|(.*?)\|\}", t, re.S)[:3]
        out[name] = [dump_planes(d) for d in dumps]
    return {"source": "tools/src/planar_444.ml:139-249", "cases": out}


def g7_packed422():
    """tools/src/packed_422.ml:56-104 expect block: a 4x4 4:2:2 frame, its YUY2 packing, and the frame unpacked again."""
    s = read("tools/src/packed_422.ml")
    t = s[s.index('let%expect_test "planar <-> packed"'):]
    rows = dump_planes(re.findall(r"\{\|(.*?)\|\}", t, re.S)[0])
    assert [len(r) for r in rows] == [4] * 4 + [2] * 8 + [8] * 4 + [4] * 4 + [2] * 8
    return {"source": "tools/src/packed_422.ml:56-104", "format": "yuy2", "frame": rows[:12], "packed": rows[12:16],
            "unpacked": rows[16:]}


def g8_header():
    """jpeg/model/test/test_encode_headers.ml:17-134: hexdump of write_headers
    c420 480x320 quality 20."""
    s = read("jpeg/model/test/test_encode_headers.ml")
    t = s[s.index("(buffer"):]
    data = bytearray()
    for m in re.finditer(r'"([0-9a-f]{8})  ((?:[0-9a-f]{2}\s+)+)\|', t):
        data += bytes(int(x, 16) for x in m.group(2).split())
    return {"source": "jpeg/model/test/test_encode_headers.ml:17-134", "width": 480, "height": 320,
            "quality": 20, "chroma": 420, "hex": data.hex()}


def g8_codewords():
    """jpeg/model/test/test_encode_codewords.ml:10-77: Encoder.size on range
    bounds, and size/magnitude/decoder-mag round trip of -15..15."""
    s = read("jpeg/model/test/test_encode_codewords.ml")
    sizes = [[int(x) for x in m.groups()] for m in re.finditer(
        r"\(\(i (\d+)\) \(lo (\d+)\) \(hi (\d+)\) \(size_lo (\d+)\) \(size_hi (\d+)\)\)", s)]
    mags = [[int(x) for x in m.groups()] for m in re.finditer(
        r"\(\(value (-?\d+)\) \(size (\d+)\) \(emag (\d+)\) \(dmag (-?\d+)\)\)", s)]
    assert len(sizes) == 12 and len(mags) == 31
    return {"source": "jpeg/model/test/test_encode_codewords.ml:10-77",
            "size_ranges_i_lo_hi_sizelo_sizehi": sizes, "value_size_emag_dmag": mags}


def g8_rle():
    """jpeg/model/test/test_rle.ml: each named case sets quant.(k) <- v on a zero
    block and prints the (run, value) list."""
    s = read("jpeg/model/test/test_rle.ml")
    cases = []
    for m in re.finditer(r'let%expect_test "([^"]+)" =(.*?)\[%expect\s*\{\|(.*?)\|\}\]', s, re.S):
        body, exp = m.group(2), m.group(3)
        if "block.rle" not in exp:
            continue
        sets = [[int(a), int(b)] for a, b in re.findall(r"block\.quant\.\((\d+)\) <- (-?\d+)", body)]
        rle = [[int(a), int(b)] for a, b in re.findall(r"\(\(run (\d+)\) \(value (-?\d+)\)\)", exp)]
        cases.append({"name": m.group(1), "set": sets, "rle": rle})
    assert len(cases) >= 9, len(cases)
    return {"source": "jpeg/model/test/test_rle.ml", "cases": cases}


def g8_code_tables():
    """jpeg/model/test/test_tables.ml:4-395: Tables.Encoder.dc_table / ac_table of the four default specifications
    (Annex K), every symbol -- length / bits / data.  dc: [category] -> [length, bits, data]; ac: [run][size] ->
    [length, bits, run, size] (rows that lack a size-0 symbol carry the model's placeholder (0, 0, run 0, size 0))."""
    s = read("jpeg/model/test/test_tables.ml")
    out = {"source": "jpeg/model/test/test_tables.ml:4-395"}
    for name in ("dc_luma", "dc_chroma"):
        t = s[s.index('("Tables.Encoder.dc_table Tables.Default.%s"' % name):]
        t = t[:t.index("|}]")]
        out[name] = [[int(x) for x in m.groups()] for m in re.finditer(r"\(\(length (\d+)\) \(bits (\d+)\) \(data (\d+)\)\)", t)]
        assert len(out[name]) == 12, (name, len(out[name]))
    for name in ("ac_luma", "ac_chroma"):
        t = s[s.index('("Tables.Encoder.ac_table Tables.Default.%s"' % name):]
        t = t[:t.index("|}]")]
        entry = r"\(\(length (\d+)\) \(bits (\d+)\) \(data \(\(run (\d+)\) \(size (\d+)\)\)\)\)"
        allv = [[int(x) for x in m.groups()] for m in re.finditer(entry, t)]
        assert len(allv) == 16 * 11, (name, len(allv))
        rows = [allv[11 * r:11 * r + 11] for r in range(16)]   # the printed table: 16 rows (runs) of 11 entries (sizes 0..10)
        for r, row in enumerate(rows):   # ... each a run's symbols in size order behind a size-0 entry, real or placeholder
            assert all(e[2] == r and e[3] == k for k, e in enumerate(row) if k), (name, r)
            assert row[0][3] == 0 and (row[0][2] == r or row[0][:3] == [0, 0, 0]), (name, r)
        out[name] = rows
    return out


def mouse_header():
    """jpeg/hardcaml/test/test_codeblock_decoder.ml prints the model's parsed
    header of Mouse480.jpg (Decoder.Header.t) and the first 64 bytes of the
    extracted entropy-coded segment."""
    s = read("jpeg/hardcaml/test/test_codeblock_decoder.ml")
    t = s[s.index('("String.subo entropy_bits ~len:64"'):]
    t = t[:t.index("Signals")]
    ent = bytearray()
    for m in re.finditer(r'"([0-9a-f]{8})  ((?:[0-9a-f]{2}\s+)+)\|', t):
        ent += bytes(int(x, 16) for x in m.group(2).split())
    assert len(ent) == 64
    frame = re.search(r"\(width (\d+)\) \(height (\d+)\)\s*\(number_of_components (\d+)\)", t)
    comps = [[int(x) for x in m.groups()] for m in re.finditer(
        r"\(\(identifier (\d+)\) \(horizontal_sampling_factor (\d+)\)\s*\(vertical_sampling_factor (\d+)\) "
        r"\(quantization_table_identifier (\d+)\)\)", t)]
    qts = [{"table_identifier": int(m.group(1)), "elements": ints(m.group(2))} for m in re.finditer(
        r"\(table_identifier (\d+)\)\s*\(elements\s*\((.*?)\)\)", t, re.S)]
    hts = [{"table_class": int(m.group(1)), "destination_identifier": int(m.group(2)),
            "lengths": ints(m.group(3)), "values": ints(m.group(4))} for m in re.finditer(
        r"\(table_class (\d+)\) \(destination_identifier (\d+)\)\s*\(lengths \((.*?)\)\)\s*\(values\s*\((.*?)\)\)", t, re.S)]
    scan = [[int(x) for x in m.groups()] for m in re.finditer(
        r"\(\(selector (\d+)\) \(dc_coef_selector (\d+)\) \(ac_coef_selector (\d+)\)\)", t)]
    assert len(qts) == 2 and len(hts) == 4 and len(comps) == 3 and len(scan) == 3
    return {"source": "jpeg/hardcaml/test/test_codeblock_decoder.ml:80-190",
            "entropy_first64_hex": ent.hex(), "width": int(frame.group(1)), "height": int(frame.group(2)),
            "components_id_h_v_tq": comps, "quant_tables": qts, "huffman_tables": hts,
            "scan_selector_dc_ac": scan}


def main():
    fixtures = {
        "g1_chen_dct.json": g1_chen(), "g2_mouse480_blocks.json": g2_mouse_blocks(),
        "g4_psnr_pins.json": g4_psnr(), "g5_quant_tables.json": g5_quant(),
        "g7_upsample.json": g7_upsample(), "g7_packed422.json": g7_packed422(),
        "g8_header_c420_480x320_q20.json": g8_header(),
        "g8_codewords.json": g8_codewords(), "g8_rle.json": g8_rle(), "g8_code_tables.json": g8_code_tables(),
        "mouse480_header.json": mouse_header(),
    }
    for name, obj in fixtures.items():
        with open(os.path.join(OUT, name), "w") as f:
            json.dump(obj, f, indent=1)
            f.write("\n")
    # the reference's own test data files (jpeg/test_data, MIT licence, <= 12 KB each)
    for fn in ("mini.jpg", "Mouse480.jpg", "mini64x64.420", "mini64x64.422", "mini64x64.444"):
        shutil.copyfile(os.path.join(REF, "jpeg/test_data", fn), os.path.join(OUT, fn))
    print("wrote", len(fixtures), "json fixtures + 5 data files to", OUT)


if __name__ == "__main__":
    sys.exit(main())
