#!/usr/bin/env python
"""Kernel-trace statistics of scripts/aux_kernels_loop.py -> profiles/<round>_aux_kernels_summary.json
usage: python scripts/summarize_aux_prof.py <rocprofv3 output dir> <out.json>"""
import json
import os
import sqlite3
import sys


def main(prof_dir, out_path):
    out = {"source": os.path.basename(prof_dir.rstrip("/")), "command": "rocprofv3 --kernel-trace --stats -- python scripts/aux_kernels_loop.py"}
    for ln in open(os.path.join(prof_dir, "run.log")):
        if ln.startswith("{"):
            out["run_under_trace"] = json.loads(ln)
    con = sqlite3.connect(os.path.join(prof_dir, "trace", "trace_results.db"))
    out["kernel_trace_stats"] = [dict(name=r[0][:200], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4])
                                 for r in con.execute("select * from top_kernels")]
    json.dump(out, open(out_path, "w"), indent=1)
    for k in out["kernel_trace_stats"]:
        print("%-90s %5d x %10.1f us" % (k["name"][:90], k["calls"], k["avg_us"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
