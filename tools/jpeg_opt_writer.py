"""Baseline JPEG files with PER-FILE OPTIMISED Huffman tables -- an input generator for tests and benchmarks (the
model's encoder, and this library's, only write the default tables; libjpeg -optimize, cameras and most web encoders
write tables fitted to each file).  Pure Python + numpy, no dependency on the library or on the test oracle: meant for
small frames and a handful of 1080p ones (about 3 s each)."""
import numpy as np


def _optimal_lengths(freq):
    """Code lengths (<= 16) for the symbols with freq > 0: ITU-T T.81 Annex K.2 (figures K.1-K.3), with the
    reserved 257th symbol so that no code is all ones.  Returns (bits[1..16] as a list of 17, huffval)."""
    f = list(freq) + [1]
    codesize = [0] * 257
    others = [-1] * 257
    while True:
        c1, v = -1, 1 << 62
        for i in range(257):
            if f[i] and f[i] <= v:
                v, c1 = f[i], i
        c2, v = -1, 1 << 62
        for i in range(257):
            if f[i] and f[i] <= v and i != c1:
                v, c2 = f[i], i
        if c2 < 0:
            break
        f[c1] += f[c2]
        f[c2] = 0
        codesize[c1] += 1
        while others[c1] >= 0:
            c1 = others[c1]
            codesize[c1] += 1
        others[c1] = c2
        codesize[c2] += 1
        while others[c2] >= 0:
            c2 = others[c2]
            codesize[c2] += 1
    bits = [0] * 33
    for i in range(257):
        if codesize[i]:
            bits[codesize[i]] += 1
    for i in range(32, 16, -1):
        while bits[i] > 0:
            j = i - 2
            while bits[j] == 0:
                j -= 1
            bits[i] -= 2
            bits[i - 1] += 1
            bits[j + 1] += 2
            bits[j] -= 1
    i = 16
    while bits[i] == 0:
        i -= 1
    bits[i] -= 1  # the reserved symbol
    huffval = [j for i in range(1, 33) for j in range(256) if codesize[j] == i]
    return bits[:17], huffval


def _canonical(bits, huffval):
    code, k, out = 0, 0, {}
    for ln in range(1, 17):
        for _ in range(bits[ln]):
            out[huffval[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return out


def _cat(v):
    return int(abs(int(v))).bit_length()


def _many_prefix_lengths(freq):
    """An AC table (valid, incomplete prefix code) whose codes of more than 10 bits sit under MORE 10-bit prefixes than
    the GPU reader has sub-tables for (csrc/hvc_hdec.h HVC_HD_SUBTABLES = 8): the three most frequent symbols get 1, 2
    and 3 bits, all 159 other run/size symbols 14 bits -- 16 codes per 10-bit prefix, ten prefixes.  Among the 14-bit
    codes the FREQUENT symbols come last, i.e. under the ninth and tenth prefix, so that a stream really uses them."""
    all_syms = [0x00, 0xF0] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    assert all(f == 0 or sym in all_syms for sym, f in enumerate(freq)), "a symbol outside the baseline alphabet"
    by_freq = sorted(all_syms, key=lambda sym: (-freq[sym], sym))
    head, tail = by_freq[:3], sorted(by_freq[3:], key=lambda sym: (freq[sym], sym))
    bits = [0] * 17
    bits[1] = bits[2] = bits[3] = 1
    bits[14] = len(tail)
    return bits, head + tail


def long_prefixes(bits, huffval):
    """the 10-bit prefixes under which a table has codes of 11..16 bits, in canonical order"""
    out = []
    for sym, (code, ln) in sorted(_canonical(bits, huffval).items(), key=lambda kv: (kv[1][1], kv[1][0])):
        if ln > 10 and (code >> (ln - 10)) not in out:
            out.append(code >> (ln - 10))
    return out


def jpeg_optimised_tables(w, h, chroma, qtabs, coefs, table_sets=2, ac_shape=None, stats=None, restart_interval=0):
    """coefs: one frame's coefficient record in the C ABI's layout (int16, component planes back to back,
    zig-zag, DC absolute) for a w x h frame of the given sampling (the encoder's geometry, encoder.ml:437-472)
    -> a baseline JPEG whose Huffman tables are the optimal ones FOR THIS FILE (table_sets = 2: luma / chroma
    pairs as every common encoder writes them; 3: one pair per component; 1: one pair for all)."""
    # (ac_shape / stats: below)
    # restart_interval = Ri > 0: a DRI segment, and behind every Ri MCUs (but the last) the bits are padded with ones to a
    # byte, an RSTm marker (m = 0 ... 7 in turn) follows and the DC predictors start again at zero (ITU-T T.81 E.1.4, B.2.4.4)
    # chroma may also be a list of (h, v) sampling factors, one per component (1..4 components, any factors 1..4): the
    # DECODER's geometry then (Decoder.init, decoder.ml:294-345: every plane = the frame rounded up to whole MCUs, scaled
    # by the component's share of the largest factors) -- samplings the model's encoder never writes but its decoder reads
    if isinstance(chroma, int):
        hs, vs = {420: (2, 2), 422: (2, 2), 444: (1, 1)}[chroma]
        ch, cv = {420: (1, 1), 422: (1, 2), 444: (1, 1)}[chroma]  # Parameters.c422 = C 1x2 (sic), encoder.ml:347-349
        comps = [(1, hs, vs, 0), (2, ch, cv, 1), (3, ch, cv, 1)]
    else:
        comps = [(i + 1, hh, vv, 0 if i == 0 else 1) for i, (hh, vv) in enumerate(chroma)]
        hs, vs = max(c[1] for c in comps), max(c[2] for c in comps)
    r_up = lambda x, m: (x + m - 1) // m * m
    Wr, Hr = r_up(w, 8 * hs), r_up(h, 8 * vs)
    dims = [(Wr * ch_ // hs // 8, Hr * cv_ // vs // 8) for _, ch_, cv_, _ in comps]  # (bw, bh) per component
    offs, at = [], 0
    for bw, bh in dims:
        offs.append(at)
        at += bw * bh * 64
    coefs = np.asarray(coefs).reshape(-1)
    assert coefs.size == at, (coefs.size, at)
    tset = {1: [0] * len(comps), 2: [0] + [1] * (len(comps) - 1), 3: list(range(len(comps)))}[table_sets]
    # pass 1: symbols in scan order (decode_seq order, decoder.ml:374-395)
    syms = []  # (table set, is_ac, symbol, extra value, extra bits); (-1, m, ...) = the marker RSTm
    pred = [0] * len(comps)
    n_mcu, mcus = (Hr // (8 * vs)) * (Wr // (8 * hs)), 0
    for my in range(Hr // (8 * vs)):
        for mx in range(Wr // (8 * hs)):
            if restart_interval and mcus and mcus % restart_interval == 0:
                syms.append((-1, (mcus // restart_interval - 1) % 8, 0, 0, 0))
                pred = [0] * len(comps)
            mcus += 1
            for ci, (_, hh, vv, _) in enumerate(comps):
                bw = dims[ci][0]
                for sy in range(vv):
                    for sx in range(hh):
                        b = coefs[offs[ci] + ((my * vv + sy) * bw + mx * hh + sx) * 64:][:64]
                        d = int(b[0]) - pred[ci]
                        pred[ci] = int(b[0])
                        s = _cat(d)
                        syms.append((tset[ci], 0, s, (d if d >= 0 else d - 1) & ((1 << s) - 1), s))
                        nz = np.flatnonzero(b[1:]) + 1
                        prev = 0
                        for k in nz:
                            run = int(k) - prev - 1
                            prev = int(k)
                            while run >= 16:
                                syms.append((tset[ci], 1, 0xF0, 0, 0))
                                run -= 16
                            v = int(b[k])
                            s = _cat(v)
                            syms.append((tset[ci], 1, (run << 4) | s, (v if v >= 0 else v - 1) & ((1 << s) - 1), s))
                        if prev != 63:
                            syms.append((tset[ci], 1, 0x00, 0, 0))
    n_sets = max(tset) + 1
    freq = [[[0] * 256 for _ in range(2)] for _ in range(n_sets)]
    for ts, ac, sym, _, _ in syms:
        if ts >= 0:
            freq[ts][ac][sym] += 1
    # ac_shape = "many_prefixes": see _many_prefix_lengths (tests of the GPU reader's overflow search); stats (a dict)
    # receives, per table set, how many coded AC symbols sit under a prefix beyond the eighth
    specs = [[_many_prefix_lengths(freq[ts][ac]) if (ac and ac_shape == "many_prefixes") else _optimal_lengths(freq[ts][ac])
              for ac in range(2)] for ts in range(n_sets)]
    if stats is not None:
        stats["symbols_beyond_eight_prefixes"] = []
        for ts in range(n_sets):
            bits_, vals_ = specs[ts][1]
            late = set(long_prefixes(bits_, vals_)[8:])
            cd = _canonical(bits_, vals_)
            stats["symbols_beyond_eight_prefixes"].append(
                sum(f for sym, f in enumerate(freq[ts][1]) if f and cd[sym][1] > 10 and (cd[sym][0] >> (cd[sym][1] - 10)) in late))
    codes = [[_canonical(*specs[ts][ac]) for ac in range(2)] for ts in range(n_sets)]
    # pass 2: the bits
    acc, nb, out = 0, 0, bytearray()
    for ts, ac, sym, extra, eb in syms:
        if ts < 0:  # RSTm: pad to a byte with ones, then the marker
            if nb:
                byte = ((acc << (8 - nb)) | ((1 << (8 - nb)) - 1)) & 0xFF
                out.append(byte)
                if byte == 0xFF:
                    out.append(0)
            acc, nb = 0, 0
            out += bytes([0xFF, 0xD0 + ac])
            continue
        c, ln = codes[ts][ac][sym]
        acc = (acc << (ln + eb)) | (c << eb) | extra
        nb += ln + eb
        while nb >= 8:
            byte = (acc >> (nb - 8)) & 0xFF
            out.append(byte)
            if byte == 0xFF:
                out.append(0)
            nb -= 8
        acc &= (1 << nb) - 1
    if nb:
        byte = ((acc << (8 - nb)) | ((1 << (8 - nb)) - 1)) & 0xFF
        out.append(byte)
        if byte == 0xFF:
            out.append(0)
    seg = lambda m, body: bytes([0xFF, m]) + (len(body) + 2).to_bytes(2, "big") + bytes(body)
    hdr = bytearray(b"\xff\xd8")
    qtabs = np.asarray(qtabs).reshape(-1, 64)
    for t in range(2):
        hdr += seg(0xDB, bytes([t]) + bytes(int(x) for x in qtabs[t]))
    hdr += seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([len(comps)]) +
               b"".join(bytes([cid, (hh << 4) | vv, tq]) for cid, hh, vv, tq in comps))
    for ts in range(n_sets):
        for ac in range(2):
            bits, vals = specs[ts][ac]
            hdr += seg(0xC4, bytes([(ac << 4) | ts]) + bytes(bits[1:17]) + bytes(vals))
    if restart_interval:
        hdr += seg(0xDD, int(restart_interval).to_bytes(2, "big"))
    hdr += seg(0xDA, bytes([len(comps)]) + b"".join(bytes([cid, (tset[i] << 4) | tset[i]]) for i, (cid, _, _, _) in enumerate(comps)) +
               bytes([0, 63, 0]))
    return bytes(hdr) + bytes(out) + b"\xff\xd9"
