#!/bin/bash
# WRITE_SIZE / write-request counts of the feature kernel with partial (4) or whole (8) chunk stores dropped
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/wr_ab; mkdir -p $out
for dbg in ${DBGS:-0 4 8 12}; do
  PAA_KERNEL_DEBUG=$dbg rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $out/d$dbg -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/run$dbg.log 2>&1
done
python - <<'PY'
import os, sqlite3, glob
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "wr_ab")
for d in (0, 4, 8, 12):
  if os.path.isdir(os.path.join(out, "d%d" % d)):
    for db in glob.glob(os.path.join(out, "d%d" % d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        r = dict((a, b) for a, b in con.execute("select counter_name, avg(counter_value) from pmc_events where name like '%st_fast%' group by counter_name"))
        n64 = r.get("TCC_EA0_WRREQ_64B_sum", 0); n = r.get("TCC_EA0_WRREQ_sum", 0)
        print("debug", d, "req", n, "req64", n64, "MB", (n64 * 64 + (n - n64) * 32) / 1e6)
PY
