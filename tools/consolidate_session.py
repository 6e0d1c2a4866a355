#!/usr/bin/env python3
"""One measurement session = a handful of files: every JSON line a session's commands printed goes into
<dir>/<tag>_lines.jsonl (each line tagged with the file it came from: "what"), non-empty stderr captures into
<dir>/<tag>_stderr.txt; text outputs (rocprofv3 summaries, microbenchmarks, timelines) stay as they are.

    python tools/consolidate_session.py profiles r03x        # merges profiles/r03x_*.json / *.jsonl / *.err in place
    python tools/consolidate_session.py gpurun_out m r04x    # a fresh session: gpurun_out/m_* -> gpurun_out/r04x_*"""
import glob
import json
import os
import sys


def main():
    d, tag = sys.argv[1], sys.argv[2]
    out_tag = sys.argv[3] if len(sys.argv) > 3 else tag
    lines, errs, gone = [], [], []
    for p in sorted(glob.glob(os.path.join(d, tag + "_*"))):
        name = os.path.basename(p)[len(tag) + 1:]
        if name in ("lines.jsonl", "stderr.txt") and out_tag == tag:
            continue
        if p.endswith(".json") or p.endswith(".jsonl"):
            for ln in open(p, errors="replace").read().splitlines():
                ln = ln.strip()
                if not ln:
                    continue
                try:
                    obj = json.loads(ln)
                except ValueError:
                    obj = {"unparsed": ln}
                lines.append(json.dumps({"what": os.path.splitext(name)[0], **(obj if isinstance(obj, dict) else {"value": obj})}))
            gone.append(p)
        elif p.endswith(".err"):
            txt = open(p, errors="replace").read().strip()
            if txt:
                errs.append("== %s\n%s\n" % (name, txt[-4000:]))
            gone.append(p)
        elif out_tag != tag:
            os.rename(p, os.path.join(d, out_tag + "_" + name))
    if lines:
        with open(os.path.join(d, out_tag + "_lines.jsonl"), "a") as f:
            f.write("\n".join(lines) + "\n")
    if errs:
        with open(os.path.join(d, out_tag + "_stderr.txt"), "a") as f:
            f.write("\n".join(errs))
    for p in gone:
        os.remove(p)
    print("%s: %d lines, %d stderr captures, %d files merged" % (out_tag, len(lines), len(errs), len(gone)))


if __name__ == "__main__":
    main()
