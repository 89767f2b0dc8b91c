#!/bin/bash
# rocprofv3 kernel trace of scripts/bench_similarity.py (run on the GPU box through gpurun)
tag=${1:-r01sim}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python scripts/bench_similarity.py 8192 > $out/bench_trace.log 2>&1
python - <<PY
import sqlite3, json
con = sqlite3.connect("$out/trace/trace_results.db")
rows = [dict(name=r[0], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in con.execute("select * from top_kernels")]
json.dump({"kernel_trace_stats": rows, "bench_line_under_trace": open("$out/bench_trace.log").read().strip().splitlines()[-1]}, open("$out/summary.json", "w"), indent=1)
for r in rows: print("%-60s calls %4d avg %.1f us" % (r["name"][:60], r["calls"], r["avg_us"]))
PY
