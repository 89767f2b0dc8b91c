#!/bin/bash
# round 6, GPU pass x: kernel trace + counter passes of one case (default big_44100) on the current build, short summary to stdout
export TMPDIR=/tmp
tag=${1:-r06x}; shift
out=gpurun_out/$tag; mkdir -p $out
for c in ${@:-big_44100}; do
  timeout 400 bash scripts/profile_kernel.sh $tag $c 20 > $out/prof_$c.log 2>&1
  python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_${c}_summary.json"))
print("$c", d["run_under_trace"]["kernel"], "%.4f ms/step  %.3e frames/s" % (d["run_under_trace"]["ms_per_step"], d["run_under_trace"]["frames_per_s"]))
for k in d["kernel_trace_stats"][:6]:
    print("   %-70s %9.1f us x %d" % (k["name"][:70], k["avg_us"], k["calls"]))
print("   traffic", d.get("traffic", {}).get("traffic_over_algorithmic"), "conflicts", d.get("lds_bank_conflict_ratio"), "issue", d.get("valu_issue_fraction"))
print("   ", {k: int(v["per_dispatch"]) for k, v in d["pmc"].items()})
PY
done 2>&1 | tee $out/summary.txt
