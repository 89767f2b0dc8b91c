#!/bin/bash
# WRITE_SIZE calibration (run on the GPU box): scripts/microbench/mb3.out under counter passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/mb3; mkdir -p $out
rocprofv3 --pmc WRITE_SIZE -d $out/p1 -o pmc -- scripts/microbench/mb3.out > $out/run1.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $out/p2 -o pmc -- scripts/microbench/mb3.out > $out/run2.log 2>&1
if [ "$1" = "full" ]; then
rocprofv3 --pmc FETCH_SIZE -d $out/p3 -o pmc -- scripts/microbench/mb3.out > $out/run3.log 2>&1
rocprofv3 --kernel-trace --stats -d $out/p4 -o trace -- scripts/microbench/mb3.out > $out/run4.log 2>&1
fi
grep "known bytes" $out/run1.log
python - <<'PY'
import os, sqlite3, glob
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "mb3")
for p in ("p1", "p2", "p3"):
    for db in glob.glob(os.path.join(out, p, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for r in con.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
            print(p, r[0][:60], r[1], "%.1f" % r[2], r[3])
for db in glob.glob(os.path.join(out, "p4", "**", "*.db"), recursive=True):
    con = sqlite3.connect(db)
    for r in con.execute("select * from top_kernels"):
        print("p4", r[0][:60], r[1:4])
PY
