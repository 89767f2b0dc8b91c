#!/bin/bash
# round 5, GPU pass aj: two A/Bs on one box, alternating builds --
#  (1) statistics pass with chunks of up to 131 072 samples (two workgroups per CU for the one-hour clip instead of four): kernel trace
#  (2) balanced runs (lib_plan.hpp: balanced_runs) for the 2 RA RB family: loops of its bench shapes
out=gpurun_out/r05aj; mkdir -p $out
export TMPDIR=/tmp
for lib in default chunk128k; do
  if [ $lib = default ]; then L=$PWD/pyaudioanalysis_amd/libpaa_hip.so; else L=$PWD/pyaudioanalysis_amd/libpaa_hip_$lib.so; fi
  PAA_HIP_LIBRARY=$L rocprofv3 --kernel-trace --stats -d $out/trace_$lib -o trace -- python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras > $out/bench_$lib.log 2>&1
  python - $out/trace_$lib $lib <<'PY'
import sqlite3, sys, os
con = sqlite3.connect(os.path.join(sys.argv[1], "trace_results.db"))
for r in list(con.execute("select * from top_kernels"))[:3]: print(sys.argv[2], r[0][:60], r[1], '%.3f us' % r[3])
PY
  rm -rf $out/trace_$lib
done
for i in 1 2; do
for c in ct_640 ct_640_spectrogram ct_800_f64 ct_800_stereo ct_400; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "equal", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_ctbal.so timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "balanced", /' >> $out/loops.jsonl 2>> $out/loops.err
done
done
python - <<'PY'
import json, collections
r = collections.OrderedDict()
for ln in open('gpurun_out/r05aj/loops.jsonl'):
    d = json.loads(ln); r.setdefault(d['case'], {}).setdefault(d['lib'], []).append(d['ms_per_step'])
for c, v in r.items():
    a, b = sum(v['balanced']) / len(v['balanced']), sum(v['equal']) / len(v['equal'])
    print(c, 'balanced', ['%.4f' % x for x in v['balanced']], 'equal', ['%.4f' % x for x in v['equal']], '%+.1f %%' % (100 * (a / b - 1)))
PY
tail -3 $out/loops.err
