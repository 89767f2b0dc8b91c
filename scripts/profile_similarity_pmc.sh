#!/bin/bash
# rocprofv3 counter passes of scripts/bench_similarity.py (run on the GPU box through gpurun): HBM traffic and issue
# statistics of sim_gram / thumb_diag / thumb_fill, one JSON summary
tag=${1:-r02simpmc}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --pmc FETCH_SIZE -d $out/p1 -o pmc -- python scripts/bench_similarity.py 8192 > $out/run1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/p2 -o pmc -- python scripts/bench_similarity.py 8192 > $out/run2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p3 -o pmc -- python scripts/bench_similarity.py 8192 > $out/run3.log 2>&1
python - <<PY
import glob, json, os, sqlite3
out = "$out"
res = {}
for p in ("p1", "p2", "p3"):
    for db in glob.glob(os.path.join(out, p, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, counter, avg, n in con.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events group by name, counter_name"):
            if "sim_" in name or "thumb_" in name:
                res.setdefault(name.split("(")[0], {})[counter] = avg
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, v in res.items():
    print(k, {c: round(x, 1) for c, x in v.items()})
PY
