#!/usr/bin/env python3
"""Markdown tables of DESIGN.md section 5, generated from the tracked records of the round (profiles/bench_<round>_n1_local.json:
the bench line of the consolidation GPU pass; profiles/<round>_<case>_summary.json: rocprofv3 kernel trace + counter passes of
scripts/profile_kernel.sh).  usage: python scripts/design_tables.py r05"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(rnd):
    # (round 6: the printed line is trimmed to fit the driver's 8 KB tail; the full record of the same run sits beside it)
    full = os.path.join(ROOT, "profiles", "bench_%s_n1_full.json" % rnd)
    line = json.load(open(full if os.path.exists(full) else os.path.join(ROOT, "profiles", "bench_%s_n1_local.json" % rnd)))
    others = line["config"]["others"]
    print("| config | kernel | step (ms) | frames/s | algorithmic GB/s (fraction of 8 TB/s) |")
    print("|---|---|---|---|---|")
    print("| cfg2 (headline) | `%s` | %.4f | %.3g | %.0f (%.3f) |" % (line["roofline"]["kernel"], line["ms_per_step"], line["value"],
                                                                  line["roofline"]["achieved"], line["roofline"]["frac"]))
    for key, e in others.items():
        if isinstance(e, dict) and "frames_per_s" in e:
            print("| %s | `%s` | %.4f | %.3g | %.0f (%.3f) |" % (key, e["kernel"], e["ms_per_step"], e["frames_per_s"], e["achieved_GBps"], e["hbm_frac"]))
    print()
    print("| case (scripts/kernel_loop.py) | kernel | kernel (us, rocprofv3) | step (ms) | HBM traffic / algorithmic | LDS conflict ratio | VALU issue |")
    print("|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "%s_*_summary.json" % rnd))):
        d = json.load(open(f))
        r = d.get("run_under_trace")
        if not r or "case" not in r or "kernel" not in r:
            continue
        tr = d.get("traffic", {})
        fmt = lambda v, s: (s % v) if v else "-"
        kernel, issue = r["kernel"], d.get("valu_issue_fraction")
        if r["case"] == "mid_stats":            # the loop runs the short-term plan first; the profiled kernel is mid_stats_kernel (a stream:
            kernel, issue = "mid_stats_kernel", None          # the issue fraction is not a meaningful figure for it)
        print("| %s | `%s` | %s | %.4f | %s | %s | %s |" % (r["case"], kernel, fmt(d.get("kernel_avg_us"), "%.1f"), r["ms_per_step"],
                                                     fmt(tr.get("traffic_over_algorithmic"), "%.2f"),
                                                     fmt(d.get("lds_bank_conflict_ratio"), "%.2f"), fmt(issue, "%.2f")))
    aux = os.path.join(ROOT, "profiles", "%s_aux_kernels_summary.json" % rnd)
    if os.path.exists(aux):
        d = json.load(open(aux))
        print()
        print("| kernel (scripts/aux_kernels_loop.py, kernel trace only) | calls | mean (us) |")
        print("|---|---|---|")
        for k in d["kernel_trace_stats"]:
            n = k["name"].split("(")[0].replace("void ", "").replace("paa::", "")
            if any(t in n for t in ("beat_kernel", "big_", "chroma_tail", "expand_deltas", "svm_binary", "wg_delta")):
                print("| `%s` | %d | %.1f |" % (n[:60], k["calls"], k["avg_us"]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r05")
