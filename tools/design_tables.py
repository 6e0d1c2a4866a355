#!/usr/bin/env python3
"""DESIGN.md's two measured tables, regenerated from ONE session's files under profiles/ so that every figure carries the
`profiles/<file>:<line>` it was read from:

    python tools/design_tables.py r05f            # rewrites the text between the KERNELS / SESSION markers of DESIGN.md

Reads profiles/<tag>_lines.jsonl (tools/gpu_round_measure.sh PART=a: one JSON line per command, tagged "what"),
profiles/traffic.json (the session's PMC entries) and profiles/<tag>_*_rocprofv3.txt (kernel trace averages)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag):
    path = os.path.join(ROOT, "profiles", tag + "_lines.jsonl")
    recs = {}
    with open(path) as f:
        for n, ln in enumerate(f, 1):
            r = json.loads(ln)
            recs[r["what"]] = (r, "`profiles/%s_lines.jsonl:%d`" % (tag, n))
    return recs


def traffic(tag, kernel, config):
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        for e in json.load(f)["entries"]:
            if e.get("session") == tag and str(e["kernel"]).startswith(kernel) and e.get("config") == config:
                return e
    return None


def trace_avg(tag, name):
    """(text, source) of the 'dominant kernel ... TIMED dispatches' or '... without the first' line of a rocprofv3 summary"""
    path = os.path.join(ROOT, "profiles", "%s_%s_rocprofv3.txt" % (tag, name))
    if not os.path.exists(path):
        return None
    best = None
    with open(path) as f:
        for n, ln in enumerate(f, 1):
            m = re.search(r"TIMED dispatches.*avg ([0-9.]+) ns", ln) or re.search(r"without the first \d+ .*avg ([0-9.]+) ns", ln)
            if m and (best is None or "TIMED" in ln):
                best = (float(m.group(1)) / 1e6, "`profiles/%s_%s_rocprofv3.txt:%d`" % (tag, name, n))
    return best


def pct(x):
    return "%.1f %%" % (100.0 * x)


def ratio(tag, kernel, config, algo):
    e = traffic(tag, kernel, config)
    return "%.4f (`profiles/traffic.json`, session %s)" % (e["hbm_bytes"] / algo, tag) if e else "—"


def tables(tag):
    R = load(tag)
    b, bsrc = R["bench"]
    o = b["others"]
    c4, c4src = R["bench_c4"]
    c3, c3src = R["bench_c3"]
    c5, c5src = R["bench_c5"]
    q16, q16src = R["bench_q16"]
    f4096, f4096src = R["bench_f4096"]
    k = []
    k.append("| kernel | computes | algorithmic bytes / unit | %% of 8 TB/s, session `%s` | counter traffic ÷ algorithmic |" % tag)
    k.append("|---|---|---|---|---|")
    rl = b["roofline"]
    k.append("| K1 `k_decode_packed`, config 2 (1024 × 1080p 4:2:0) | a1 – a6 | 192 B / block (128 in + 64 out) | **%s** (%.4f ms, %s) | %s |"
             % (pct(rl["frac"]), rl["kernel_ms"], bsrc, ratio(tag, "k_decode_packed", 2, rl["algorithmic_bytes_per_launch"])))
    r4 = c4["roofline"]
    k.append("| K1 on config 4's launch (128 × 4K 4:4:4) | same | same | **%s** (%.4f ms, %s) | %s |"
             % (pct(r4["frac"]), r4["kernel_ms"], c4src, ratio(tag, "k_decode_packed", 4, r4["algorithmic_bytes_per_launch"])))
    r5 = c5["roofline"]
    k.append("| K3 `k_encode`, config 5 (256 × 4K 4:2:0) | a12 – a14 | 192 B / block (64 in + 128 out) | **%s** (%.4f ms, %s) | %s |"
             % (pct(r5["frac"]), r5["kernel_ms"], c5src, ratio(tag, "k_encode", 5, r5["algorithmic_bytes_per_launch"])))
    f = o["fused444_512x1080p"]
    k.append("| fused `k_decode_444` (+ `k_reinterp_444`), 512 × 1080p | a1 – a6 + a9 crop + a11 | 128 B / needed block + 3·W·H out | **%s** (%.4f ms, `others` of %s) | %s |"
             % (pct(f["frac_of_8TBps"]), f["ms"], bsrc, ratio(tag, "k_decode_444", 7, f["algorithmic_bytes_per_launch"])))
    k2, s2 = o["k2_upsample420_512_planes"], o["subsample420_512_planes"]
    k.append("| K2 `k_upsample420_x8`, 512 planes 960 × 540 → 1920 × 1080 | a11 | 5 B / source sample | %s (`others`; a 0.2 ms kernel) | — |" % pct(k2["frac_of_8TBps"]))
    k.append("| `k_subsample420` (`hvc_yuv.hip`), 512 planes 1920 × 1080 → 960 × 540 | `subsample_hv2` | 5 B / destination sample | %s (`others`) | — |" % pct(s2["frac_of_8TBps"]))
    k.append("| `k_decode_q16` (north star's mapping; A/B only) | a1 – a6 | 192 B / block | %s (%s) | — |" % (pct(q16["roofline"]["frac"]), q16src))
    if "wide" in R:
        wd, wdsrc = R["wide"]
        k.append("| `k_decode_wide_all` (every block in the model's 63-bit arithmetic: a 16-bit DQT, `hvc_set_decode_kernel(ctx, 2)`), 64 × 1080p | a1 – a6 | 192 B / block | **%s** (%.4f ms, %s); VALU-bound | %s |"
                 % (pct(wd["frac_of_8TBps"]), wd["kernel_ms"], wdsrc, ratio(tag, "k_decode_wide_all", 13, wd["all_blocks"] * 192)))

    t = []
    t.append("| what | measured | source |")
    t.append("|---|---|---|")
    su = b.get("sustained") or {}
    t.append("| **config 2 (headline)**: 1080p 4:2:0, 1024 frames (9.6 GB) per launch, %d timed steps | **%s Mpixel/s**; `k_decode_packed` %.4f ms = %.0f GB/s algorithmic = **%s of 8 TB/s**; K5 verified: %s; wide-path blocks: %d | %s |"
             % (b["steps"], "{:,.0f}".format(b["value"]).replace(",", " "), rl["kernel_ms"], rl["achieved"], pct(rl["frac"]), b["checksum"]["verified"],
                b["config"]["wide_path_blocks"], bsrc))
    if su:
        t.append("| … `sustained`: %.1f s of the same step after the timed steps | first decile %.4f ms, last decile %.4f ms, GPU busy %s, output unchanged: %s | same line |"
                 % (su["wall_s"], su["first_decile_ms"], su["last_decile_ms"], pct(su["gpu_busy_fraction"]), su.get("output_unchanged")))
    cb = b.get("cpu_baseline")
    if cb:
        t.append("| … CPU baseline: the oracle's scalar block stage, same frames (kind `port`) | %.0f Mpixel/s on 1 core; %.0f on %d threads | same line |"
                 % (cb["value"], cb["all_cores"]["value"], cb["all_cores"]["cores"]))
    ol = rl.get("other_layout")
    if ol and "frac" in ol:
        t.append("| … the same frames, %s (what the library's own entry points lay out), same run | `k_decode_packed` %.4f ms = %s; K5 verified: %s | same line |"
                 % (ol["layout"], ol["kernel_ms"], pct(ol["frac"]), ol["verified"]))
    t.append("| … verification | K5 over **all %d records** of the launch (record r = distinct frame r mod %d) against `tests/golden/bench_checksums.json` | same line |"
             % (b["checksum"]["frames"], b["checksum"].get("distinct", 8)))
    t.append("| … same workload, 4096 frames (38.5 GB) per call, cut into launches by the library | %s | %s |" % (pct(f4096["roofline"]["frac"]), f4096src))
    t.append("| … same workload through `k_decode_q16` | %s Mpixel/s = %s | %s |" % ("{:,.0f}".format(q16["value"]).replace(",", " "), pct(q16["roofline"]["frac"]), q16src))
    t.append("| **config 4** (`--config 4`): 4K 4:4:4, one GPU's 2048-frame shard, %s | **%s Mpixel/s**, %.2f ms per pass over the shard; kernel %.4f ms = **%s**; verified: %s | %s |"
             % ("resident" if "resident in HBM" in c4["config"]["workload"] else "one chunk re-used", "{:,.0f}".format(c4["value"]).replace(",", " "),
                c4["ms_per_step"], r4["kernel_ms"], pct(r4["frac"]), c4["checksum"]["verified"], c4src))
    h, g = c3["host_reader"], c3["gpu_reader"]
    t.append("| **config 3** (`--config 3`): 4096 × 1080p JPEG files, host Huffman (%d threads) ‖ H2D ‖ K1 | **%.2f Gpixel/s** (%.1f ms per 4096 files; GPU busy %s: host-bound); verified: %s | %s |"
             % (c3["config"]["host_threads_per_rank"], h["value"] / 1e3, h["ms_per_step"], pct(h["gpu_busy_fraction"]), h["verified"], c3src))
    if "h2d_GBps" in h:
        c3r = c3["roofline"]
        t.append("| … SURVEY §8(d) C3's figures of that line | upload %.1f GB/s; overlap fraction %.2f (1 − wall ÷ (Huffman ÷ threads + upload + kernels)); its %d-frame `k_decode_packed` launches: %.0f GB/s algorithmic, counter traffic ÷ algorithmic %s | same line |"
                 % (h["h2d_GBps"], h["overlap_fraction"], h["kernel_launch_frames"], c3r["achieved"],
                    ratio(tag, "k_decode_packed", 3, c3r["algorithmic_bytes_per_launch"])))
    t.append("| … the same files, Huffman reader on the GPU (`hvc_jpeg_decode_batch_gpu`) | **%.1f Gpixel/s** (%.1f ms; %.0f MB of unstuffed segments up instead of %.0f MB of coefficients: upload-bound); verified: %s | same line |"
             % (g["value"] / 1e3, g["ms_per_step"], g["h2d_MB_per_step"], h["h2d_MB_per_step"], g["verified"]))
    if c3.get("cpu_baseline"):
        t.append("| … CPU baseline: the oracle's `decode_a_frame` (Huffman + block stage), 1 thread | %.1f Mpixel/s | same line |" % c3["cpu_baseline"]["value"])
    if "seam" in R:
        sm, smsrc = R["seam"]
        t.append("| **the asynchronous seam** (`hvc_host_alloc`, `hvc_decode_frames_submit` / `hvc_wait`, %d slots of %d frames): 4096 × 1080p coefficient records, %d caller threads refilling the next slot's pinned record, pixels to HBM | **%.2f Gpixel/s** (%.1f ms); upload %.1f GB/s; overlap fraction %.2f; K5 over all %d records: %s | %s |"
                 % (sm["slots"], sm["frames_per_slot"], sm["host_threads"], sm["value"] / 1e3, sm["wall_ms"], sm["h2d_GBps"], sm["overlap_fraction"],
                    sm["checksum"]["records"], sm["checksum"]["verified"], smsrc))
    if "seam_host" in R:
        sh, shsrc = R["seam_host"]
        t.append("| … pixels back into pinned host slots as well (%d records, every one compared byte for byte) | **%.2f Gpixel/s**; up %.1f GB/s ‖ down %.1f GB/s; overlap fraction %.2f; verified: %s | %s |"
                 % (sh["frames"], sh["value"] / 1e3, sh["h2d_GBps"], sh["d2h_GBps"], sh["overlap_fraction"], sh["checksum"]["verified"], shsrc))
    t.append("| **config 5** (`--config 5`): 4K 4:2:0 encode (fDCT + quantise), 256 frames (9.6 GB) per launch | **%s Mpixel/s**; `k_encode` %.4f ms = **%s**; verified: %s; CPU baseline %.0f Mpixel/s on 1 core | %s |"
             % ("{:,.0f}".format(c5["value"]).replace(",", " "), r5["kernel_ms"], pct(r5["frac"]), c5["checksum"]["verified"],
                (c5.get("cpu_baseline") or {}).get("value", float("nan")), c5src))
    a, bb = o["config5_files_host_coder_256_frames"], o["config5_files_gpu_coder_256_frames"]
    t.append("| config 5 end to end (raw 4K frames → `.jpg` files), host coder / GPU coder | %.2f / **%.1f Gpixel/s** (host-bound / upload-bound), both verified byte for byte | `others` of %s |"
             % (a["Gpixel_s"], bb["Gpixel_s"], bsrc))
    t.append("| `others` of the headline line (what the driver's own run witnesses) | config 4's launch %s, K3 %s, fused 4:4:4 %s, K2 %s, `subsample_hv2` %s; config 3 host / GPU reader %.1f / %.1f Gpixel/s; all `verified` | %s |"
             % (pct(o["config4_launch_128x4K444"]["frac_of_8TBps"]), pct(o["k3_encode_256x4K420"]["frac_of_8TBps"]), pct(f["frac_of_8TBps"]),
                pct(k2["frac_of_8TBps"]), pct(s2["frac_of_8TBps"]), o["config3_host_reader_4096_files"]["Gpixel_s"],
                o["config3_gpu_reader_4096_files"]["Gpixel_s"], bsrc))
    if "c7" in R:
        c7, c7src = R["c7"]
        t.append("| next-3: 1080p 4:2:0 records → tight 4:4:4 frames, 512 frames: fused against decode + crop + 2 × K2 | %.4f ms = %s against %.4f ms (× %.2f) | %s |"
                 % (c7["fused_ms"], pct(c7["frac_of_8TBps"]), c7["separate_ms"], c7["speedup_vs_separate"], c7src))
    if "c6" in R:
        c6, c6src = R["c6"]
        t.append("| host-buffer boundary (`HVC_MEM_HOST`): PCIe-inclusive, never `value` | %.1f Gpixel/s | %s |" % (c6["value"] / 1e3, c6src))
    if "wide" in R and "wide_dqt16" in R:
        (wd, wdsrc), (w16, w16src) = R["wide"], R["wide_dqt16"]
        t.append("| every block through `k_decode_wide_all` (64 × 1080p): `hvc_set_decode_kernel(ctx, 2)` / a DQT entry of 40000 | %.0f / %.0f Gpixel/s = %s / %s; equal to the packed kernel's frames: %s / %s | %s, %s |"
                 % (wd["value"] / 1e3, w16["value"] / 1e3, pct(wd["frac_of_8TBps"]), pct(w16["frac_of_8TBps"]), wd["checksum"]["verified"],
                    w16["checksum"]["verified"], wdsrc, w16src.replace("profiles/%s_lines.jsonl" % tag, "")))
    if "fixup" in R:
        fx, fxsrc = R["fixup"]
        t.append("| the exactness contract's worst case under 8-bit tables: adversarial ± 2047 records, every block fails the packed kernel's guard (packed kernel + fix-up list + `k_decode_wide`, whole call) | %.1f Gpixel/s = %s (%.4f ms per 64 frames; %d of %d blocks through the list); equal to `k_decode_wide_all`'s frames: %s | %s |"
                 % (fx["value"] / 1e3, pct(fx["frac_of_8TBps"]), fx["ms_per_call"], fx["wide_path_blocks"], fx["all_blocks"], fx["checksum"]["verified"], fxsrc))
    if "huffman_gpu" in R:
        hg, hgsrc = R["huffman_gpu"]
        t.append("| GPU Huffman coder alone (4K 4:2:0, q75) | %.0f Gpixel/s | %s |" % (hg["value"] / 1e3, hgsrc))
    reh = []
    for what, label in (("bench_rehearsal2", "2 ranks config 2"), ("bench_rehearsal4_c4", "4 ranks config 4"), ("bench_rehearsal2_c3", "2 ranks config 3"),
                        ("bench_rehearsal2_c5", "2 ranks config 5")):
        if what in R:
            r, src = R[what]
            reh.append("%s: %s of %d verified (%s)" % (label, r["checksum"].get("ranks_verified"), r["n_gpus"], src.replace("profiles/%s_lines.jsonl" % tag, "")))
    if reh:
        t.append("| N-rank rehearsals on this one GPU (all ranks on cuda:0, gloo; never measurements) | %s | `profiles/%s_lines.jsonl`, lines as given |" % ("; ".join(reh), tag))
    prof = []
    for name, label in (("decode", "`python bench.py`, `k_decode_packed`"), ("decode_c4", "`--config 4`"), ("decode_c3", "config 3's 32-frame launches"),
                        ("encode", "`k_encode`"), ("444", "`k_decode_444`"), ("wide", "`k_decode_wide_all`")):
        ta = trace_avg(tag, name)
        if ta:
            prof.append("%s %.4f ms (%s)" % (label, ta[0], ta[1]))
    return "\n".join(k), "\n".join(t), ("trace averages — " + "; ".join(prof) + "; counter bytes: `profiles/traffic.json`, session " + tag + ".") if prof else ""


KERNEL_FILES = (("hvc_kernels.hip", ("k_decode_packed", "k_decode_wide", "k_decode_wide_all", "k_decode_q16", "k_decode_444", "k_reinterp_444", "k_encode",
                                      "k_upsample420", "k_upsample420_x8", "k_abs_error", "k_checksum")),
                ("hvc_kernels.h", ("xcd_work",)), ("hvc_yuv.hip", ("k_subsample420",)), ("hvc_hdec.hip", ("k_hd_sync", "k_hd_write2")),
                ("hvc_huff.hip", ("k_huff_len", "k_huff_emit")))


def anchors():
    """`name` file:line of every kernel DESIGN.md names: the line of its definition"""
    out = []
    for fn, names in KERNEL_FILES:
        lines = open(os.path.join(ROOT, "video-coding_amd", "csrc", fn)).read().splitlines()
        for name in names:
            pat = re.compile(r"(__global__|__device__ __forceinline__ void).*\b%s\(" % re.escape(name))
            hit = [n for n, ln in enumerate(lines, 1) if pat.search(ln)]
            if hit:
                out.append("`%s` `%s:%d`" % (name, fn, hit[0]))
    return " · ".join(out)


def main():
    tag = sys.argv[1]
    k, t, p = tables(tag)
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for name, body in (("ANCHORS", anchors()), ("KERNELS", k), ("SESSION", t), ("PROFILES", p)):
        a, b = "<!-- %s:BEGIN -->" % name, "<!-- %s:END -->" % name
        i, j = s.index(a) + len(a), s.index(b)
        s = s[:i] + "\n" + body + "\n" + s[j:]
    s = re.sub(r"one session, `[^`]*`, on", "one session, `%s`, on" % tag, s)
    open(path, "w").write(s)
    print("DESIGN.md: tables of session %s written (%d lines)" % (tag, s.count("\n")))


if __name__ == "__main__":
    main()
