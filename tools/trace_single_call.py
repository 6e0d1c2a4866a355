#!/usr/bin/env python3
"""One file at a time, the reference's own call shape (Decoder.decode_a_frame, decoder.mli:59): hvc_jpeg_decode on ONE
1080p 4:2:0 file, a few times, for a kernel trace -- and, from that trace, the timeline of one call.

    rocprofv3 --kernel-trace -d DIR -o trace -- python3 tools/trace_single_call.py --quality 3
    python3 tools/trace_single_call.py --timeline DIR        # the last call's launches: offset, duration, gap
"""
import argparse
import glob
import json
import os
import sqlite3
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeline(d):
    db = (glob.glob(os.path.join(d, "*.db")) or glob.glob(os.path.join(d, "*", "*.db")) or glob.glob(os.path.join(d, "*", "*", "*.db")))[0]
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, start, end, grid_x, workgroup_x from kernels where name like '%hvc::%' order by start"))
    # a call starts at k_hd_frame_of (the reader's first launch) and ends before the next one
    starts = [i for i, r in enumerate(rows) if "k_hd_frame_of" in r[0]]
    if not starts:
        print("no reader launches in the trace")
        return
    a = starts[-1]
    call = rows[a:]
    t0 = call[0][1]
    prev_end = t0
    busy = 0
    print("launches of the last call: offset_us duration_us gap_before_us  grid x wg  kernel")
    for name, s, e, gx, wg in call:
        short = name.split("hvc::")[-1].split("(")[0][:40]
        print("  %8.1f %8.1f %8.1f   %5d x %-4d %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, gx // max(wg, 1), wg, short))
        busy += e - s
        prev_end = e
    print("kernels: %d, first start to last end %.1f us, of which kernels ran %.1f us" % (len(call), (prev_end - t0) / 1e3, busy / 1e3))
    per = {}
    for name, s, e, gx, wg in call:
        k = name.split("hvc::")[-1].split("(")[0].split("<")[0]
        per[k] = per.get(k, [0, 0.0])
        per[k][0] += 1
        per[k][1] += (e - s) / 1e3
    for k, (n, us) in sorted(per.items(), key=lambda x: -x[1][1]):
        print("  %-18s x%-3d %8.1f us" % (k, n, us))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quality", type=int, default=3)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--timeline", default=None)
    args = ap.parse_args()
    if args.timeline:
        return timeline(args.timeline)
    import video_coding_amd as hvc
    from video_coding_amd.synth import synth_pixels
    ctx = hvc.Context(0)
    w, h = 1920, 1080
    y = synth_pixels(11, 1088, 1920)[:h]
    u = synth_pixels(12, 544, 960)[:540]
    v = synth_pixels(13, 544, 960)[:540]
    jpg = ctx.jpeg_encode(y, u, v, w, h, 420, args.quality)
    t = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        ctx.jpeg_decode(jpg)
        t.append(time.perf_counter() - t0)
    print(json.dumps({"file_kB": round(len(jpg) / 1024, 1), "hvc_jpeg_decode_ms_min": round(1e3 * min(t[2:]), 3),
                      "hvc_jpeg_decode_ms_last": round(1e3 * t[-1], 3)}))
    ctx.close()


if __name__ == "__main__":
    main()
