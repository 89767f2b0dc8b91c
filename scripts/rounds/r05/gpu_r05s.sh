#!/bin/bash
# round 5, GPU pass s: split transform with tasks handed out through a counter (pairs first), time-domain partials with four
# loads in flight: parity of the big-window tests, trace + counters of 44100/22050, loops of the workgroup-per-frame shapes
out=gpurun_out/r05s; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header --durations=8 --maxfail=20 -k "big or workgroup" 2>&1 | tail -40) > $out/tests.log
tail -3 $out/tests.log
for c in big_44100; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
python - <<'PY'
import json
for c in ("big_44100",):
    d = json.load(open('gpurun_out/r05_%s_summary.json' % c))
    print(c, d['run_under_trace']['ms_per_step'])
    for k in d['kernel_trace_stats'][:6]: print('   ', k['name'][:100], k['calls'], k['avg_us'])
    print({k: v['per_dispatch'] for k, v in d['pmc'].items()})
PY
for c in big_16000 big_16000_1h big_8000_batch; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 30 2>&1 | tail -1 | cut -c1-200
done
