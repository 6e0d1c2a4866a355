#!/usr/bin/env python3
"""Summarise the rocprofv3 (rocpd sqlite) outputs of tools/gpu_profile.sh into a
small text file for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_<tag> [untimed launches, default 10] > profiles/<name>.txt

(bench.py issues 8 setup launches + W warm-up steps before the timed ones: pass 8 + W for its profiles.)
"""
import glob
import os
import sqlite3
import sys


def q(db, sql):
    return list(sqlite3.connect(db).execute(sql))


WARMUP = 10


def main(d):
    d_ = d
    print("# rocprofv3 summary of %s" % d)
    tr = glob.glob(os.path.join(d, "trace", "*.db"))
    if tr:
        print("\n## --kernel-trace --stats  (durations in ns)")
        print("%-60s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        # (from the dispatch table itself, in ns: the database's top_kernels view reports its totals in another unit than
        # its averages, which round 3's summaries printed side by side as if they were one -- VERDICT r3)
        rows = q(tr[0], "select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc")
        grand = float(sum(r[2] for r in rows)) or 1.0
        for name, calls, total, avg in rows:
            print("%-60s %8d %14.0f %12.1f %7.2f" % (name[:60], calls, total, avg, 100.0 * total / grand))
        # the dominant hvc kernel: average over all dispatches and over the timed ones only (the first
        # `warmup` launches of a bench run are the untimed warm-ups, still off the sustained clock)
        dom = q(tr[0], "select name from top_kernels where name like '%hvc::%' order by total_duration desc limit 1")
        if dom:
            ds = [r[0] for r in q(tr[0], "select duration from kernels where name = ? order by start".replace("?", "'%s'" % dom[0][0].replace("'", "''")))]
            w = WARMUP if len(ds) > WARMUP else 0
            print("\ndominant kernel %s: %d dispatches, avg %.1f ns; without the first %d (setup / warm-up): avg %.1f ns"
                  % (dom[0][0][:60], len(ds), sum(ds) / len(ds), w, sum(ds[w:]) / len(ds[w:])))
        timed = None
        for log in ("trace_bench.log", "trace.log"):
            lp = os.path.join(d_, log)
            if os.path.exists(lp):
                for line in open(lp):
                    if line.startswith("{"):
                        print("bench line of this very run (HIP events inside the profiled process):\n  " + line.strip())
                        try:
                            import json
                            rec = json.loads(line)
                            cfg = rec.get("config", {})
                            timed = rec["steps"] * (cfg["frames_per_gpu_per_step"] // cfg["frames_per_launch"])
                        except (ValueError, KeyError, TypeError):
                            pass
        if dom and timed:  # bench.py: setup + warm-up, then exactly K timed steps, then the `sustained` run
            t = ds[WARMUP:WARMUP + timed]
            if t:
                print("dominant kernel, the %d TIMED dispatches (after %d of setup / warm-up; %d more belong to the sustained run and to `others`): avg %.1f ns"
                      % (len(t), WARMUP, max(0, len(ds) - WARMUP - timed), sum(t) / len(t)))
        # The trace's vgpr_count column is HALF the kernel's register allocation (rounded up to the granule of 8) on gfx950:
        # k_decode_packed 60 for the 116 of its code object (-> 120), k_decode_wide 108 for 216, k_decode_444 40 for 76
        # (-> 80) -- compared against the .vgpr_count of the code objects (the column read "vgpr=60" for a
        # kernel whose 116 registers are what holds it to 4 waves per SIMD).  Printed as the allocation it stands for.
        rows = q(tr[0], "select name, duration, vgpr_count, sgpr_count, grid_x, grid_y, workgroup_x from kernels "
                        "where name like '%hvc::%' order by start")
        print("\nper-dispatch (hvc kernels): name duration_ns vgpr_alloc (= 2 x the trace's vgpr_count column) sgpr grid wg")
        for r in rows[:160]:
            print("  %-48s %9d vgpr_alloc=%d sgpr=%d grid=%dx%d wg=%d" % (r[0][:48], r[1], 2 * r[2], r[3], r[4], r[5], r[6]))
        if len(rows) > 160:
            print("  ... %d more dispatches (the totals above cover them)" % (len(rows) - 160))
    print("\n## --pmc passes (average per dispatch of each hvc kernel)")
    for sub in sorted(glob.glob(os.path.join(d, "pmc_*"))):
        dbs = glob.glob(os.path.join(sub, "*.db"))
        if not dbs:
            continue
        for kn, cn, avg, n in q(dbs[0], "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                        "where kernel_name like '%hvc::%' group by kernel_name, counter_name"):
            extra = ""
            if cn == "FETCH_SIZE":
                extra = "  KB -> x1024 x2 (gfx950 16B/lane read correction) = %.1f MB" % (avg * 1024 * 2 / 1e6)
            if cn == "WRITE_SIZE":
                extra = "  KB -> x1024 = %.1f MB" % (avg * 1024 / 1e6)
            print("  %-40s %-22s %16.1f  (n=%d)%s" % (kn[:40], cn, avg, n, extra))


def traffic_json(d, kernel_substr, frames, config=2, session="", what=""):
    """HBM bytes per launch of one kernel from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE x2 for 16 B/lane streaming reads on gfx950)."""
    out = {}
    for sub, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        dbs = glob.glob(os.path.join(d, sub, "*.db"))
        rows = q(dbs[0], "select avg(value) from counters_collection where kernel_name like '%%%s%%' and "
                         "counter_name='%s'" % (kernel_substr, name))
        out[name] = rows[0][0]
    fetch = out["FETCH_SIZE"] * 1024 * 2
    write = out["WRITE_SIZE"] * 1024
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import video_coding_amd as hvc   # the library the profiled commands loaded: the pass is of THIS build's kernels
    return {"kernel": kernel_substr, "config": config, "session": session, "build": hvc.hvc.kernel_build_id(),
            "frames_per_launch": frames, "fetch_bytes": fetch,
            "write_bytes": write, "hbm_bytes": fetch + write,
            "source": os.path.basename(os.path.normpath(d)) + (": " + what if what else ""),
            "corrections": "FETCH_SIZE KB x1024 x2 (gfx950 16 B/lane read undercount), WRITE_SIZE KB x1024"}


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[2] == "--traffic":  # <dir> --traffic <kernel> <frames> [config] [session] [the profiled command]
        import json
        print(json.dumps(traffic_json(sys.argv[1], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 2,
                                      sys.argv[6] if len(sys.argv) > 6 else "", sys.argv[7] if len(sys.argv) > 7 else ""), indent=1))
    else:
        if len(sys.argv) > 2:
            WARMUP = int(sys.argv[2])
        main(sys.argv[1])
