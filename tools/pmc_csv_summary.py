#!/usr/bin/env python3
"""Average per dispatch of every counter in the rocprofv3 (rocpd sqlite) databases below the given directories, for the
hvc kernels -- one line per (directory, kernel, counter).  Used for PMC passes taken outside tools/gpu_profile.sh.

    python tools/pmc_csv_summary.py [--kernel SUBSTR] gpurun_out/pmc_<tag>_f1024 gpurun_out/pmc_<tag>_f2048 ...

Without --kernel the fix-up, encode and checksum kernels are left out (a decode session's by-catch); with it only kernels
whose name holds SUBSTR are listed.
"""
import glob
import os
import sqlite3
import sys


def main(dirs):
    only = None
    if len(dirs) > 1 and dirs[0] == "--kernel":
        only, dirs = dirs[1], dirs[2:]
    for d in dirs:
        print("## %s" % d)
        for db in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            try:
                rows = list(sqlite3.connect(db).execute(
                    "select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                    "where kernel_name like '%hvc::%' group by kernel_name, counter_name"))
            except sqlite3.Error as e:
                print("  %s: %s" % (db, e))
                continue
            for kn, cn, avg, n in rows:
                if (only not in kn) if only else ("k_decode_wide" in kn or "k_encode" in kn or "k_checksum" in kn):
                    continue
                print("  %-44s %-46s %18.1f  (n=%d)" % (kn[:44], cn, avg, n))


if __name__ == "__main__":
    main(sys.argv[1:])
