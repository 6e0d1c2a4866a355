#!/usr/bin/env python3
"""GPU work per chunk of the file-level pipeline, from a rocprofv3 kernel trace (rocpd database) of
tools/bench_configs.py --config 3 --gpu-entropy: the reader's kernels and the block stage of every chunk of the largest
size in the trace, summed per chunk and averaged.  Launches are attributed to chunks per stream: on its stream a chunk
is the run of launches from the first k_hd_sync after another kernel to the block stage.

    python tools/reader_chunk_ms.py gpurun_out/prof_<tag>/trace
"""
import glob
import os
import sqlite3
import sys

FAMS = ("k_hd_sync_tail", "k_hd_sync", "k_hd_round", "k_hd_scan", "k_hd_write2", "k_hd_write", "k_hd_dc", "k_decode_packed", "k_decode_wide")


def family(name):
    for k in FAMS:
        if k in name:
            return k
    return None


def main(d):
    db = glob.glob(os.path.join(d, "*.db")) or glob.glob(os.path.join(d, "trace", "*.db"))
    con = sqlite3.connect(db[0])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    lane = "stream_id" if "stream_id" in cols else "queue_id" if "queue_id" in cols else "0"
    rows = list(con.execute("select name, duration, grid_x, start, %s from kernels where name like '%%hvc::%%' order by start" % lane))
    chunks = []
    open_by_lane = {}
    for name, dur, grid, start, ln in rows:
        fam = family(name)
        if fam is None:
            continue
        cur = open_by_lane.get(ln)
        if fam == "k_hd_sync" and (cur is None or cur["seen_other"]):
            cur = {"seen_other": False, "subs": grid, "k": {}, "sync_rounds": []}
            open_by_lane[ln] = cur
            chunks.append(cur)
        if cur is None:
            continue
        if fam != "k_hd_sync":
            cur["seen_other"] = True
        else:
            cur["sync_rounds"].append(dur / 1e3)
        cur["k"][fam] = cur["k"].get(fam, 0.0) + dur / 1e6
    if not chunks:
        print("no reader launches in the trace")
        return
    full = max(c["subs"] for c in chunks)
    sel = [c for c in chunks if c["subs"] == full and ("k_hd_write2" in c["k"] or "k_hd_write" in c["k"])]
    # the block stage runs on the pipeline's own stream: its launches over whole chunks are the ones with the largest grid
    k1 = [(dur, gy) for name, dur, grid, start, ln in rows if "k_decode_packed" in name
          for gy in [list(con.execute("select grid_y from kernels where start = %d" % start))[0][0]]]
    if k1:
        gmax = max(g for _, g in k1)
        k1d = [dur / 1e6 for dur, g in k1 if g == gmax]
        for c in sel:
            c["k"]["k_decode_packed"] = sum(k1d) / len(k1d)
    print("chunks in the trace: %d, of the largest size (%d subsequence lanes launched): %d" % (len(chunks), full, len(sel)))
    tot = 0.0
    for fam in FAMS:
        v = [c["k"].get(fam, 0.0) for c in sel]
        if any(v):
            print("  %-16s %.3f ms" % (fam, sum(v) / len(v)))
            tot += sum(v) / len(v)
    print("  %-16s %.3f ms per chunk" % ("sum", tot))
    n = min(len(c["sync_rounds"]) for c in sel)
    print("  k_hd_sync rounds (us):", " ".join("%.0f" % (sum(c["sync_rounds"][r] for c in sel) / len(sel)) for r in range(n)))


if __name__ == "__main__":
    main(sys.argv[1])
