"""Fold the rocprofv3 (rocpd sqlite) outputs of scripts/profile.sh into one JSON summary under profiles/."""
import json
import os
import sqlite3
import sys


def main(prof_dir, out_path, kernel_like="%st_fast%"):
    out = {"source": prof_dir}
    con = sqlite3.connect(os.path.join(prof_dir, "trace", "trace_results.db"))
    out["kernel_trace_stats"] = [dict(name=r[0], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4])
                                 for r in con.execute("select * from top_kernels")]
    pm = {}
    for n in sorted(os.listdir(prof_dir)):
        db = os.path.join(prof_dir, n, "pmc_results.db")
        if not os.path.exists(db):
            continue
        con = sqlite3.connect(db)
        nd = con.execute("select count(distinct dispatch_id) from pmc_events where name like ?", (kernel_like,)).fetchone()[0]
        for r in con.execute("select counter_name, sum(counter_value) from pmc_events where name like ? "
                             "group by counter_name", (kernel_like,)):
            pm[r[0]] = {"per_dispatch": r[1] / max(nd, 1), "dispatches": nd}
    out["pmc_feature_kernel"] = pm
    for k in ("FETCH_SIZE", "WRITE_SIZE"):
        if k in pm:
            pm[k]["unit"] = "KB (rocprofv3); FETCH_SIZE under-reports wide coalesced reads 2x on gfx950"
    logs = [f for f in os.listdir(prof_dir) if f.startswith("bench_trace")]
    for f in logs:
        for line in open(os.path.join(prof_dir, f)):
            if line.startswith("{"):
                out["bench_line_under_trace"] = json.loads(line)
    # HBM traffic of the feature kernel per launch, corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE is in
    # KB and under-reports wide coalesced reads 2x; WRITE_SIZE in KB as reported) -> profiles/latest_traffic.json
    bl = out.get("bench_line_under_trace")
    if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm and bl:
        fetch = pm["FETCH_SIZE"]["per_dispatch"] * 1024.0 * 2.0
        write = pm["WRITE_SIZE"]["per_dispatch"] * 1024.0
        traffic = {"kernel": bl["config"]["kernel"], "frames": bl["config"].get("frames_per_step_rank0"),
                   "rows": bl["config"]["rows"], "fetch_bytes_corrected": fetch, "write_bytes": write,
                   "hbm_bytes_per_launch": fetch + write, "round": os.path.basename(out_path)[:3],
                   "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 on gfx950)"
                             % os.path.basename(out_path)}
        out["traffic"] = traffic
        json.dump(traffic, open(os.path.join(os.path.dirname(os.path.abspath(out_path)), "latest_traffic.json"), "w"), indent=1)
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
