#!/bin/bash
# round 5, GPU pass r: split transform with next-task prefetch, two outputs per lane in flight in the first pass, one-sweep block
# energies for rows read from L2, sixteen-wave time kernel: parity of the big-window tests + trace and counters of 44100/22050
out=gpurun_out/r05r; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header --durations=8 --maxfail=20 -k "big or workgroup" 2>&1 | tail -40) > $out/tests.log
tail -8 $out/tests.log
for c in big_44100; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
python - <<'PY'
import json
for c in ("big_44100",):
    d = json.load(open('gpurun_out/r05_%s_summary.json' % c))
    print(c, d['run_under_trace']['ms_per_step'])
    for k in d['kernel_trace_stats'][:6]: print('   ', k['name'][:100], k['calls'], k['avg_us'])
PY
