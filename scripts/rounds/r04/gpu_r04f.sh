#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
(timeout 600 python -m pytest tests -m gpu -q --no-header -x --durations=8 2>&1 | tail -30) > $out/tests.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
for c in w2400 w2205; do timeout 600 bash scripts/profile_kernel.sh r04 $c > $out/prof_$c.log 2>&1; done
tail -22 $out/tests.log; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04f/bench.json').read().strip().split('\n')[-1])
print(r['value'], r['ms_per_step'], r['roofline']['kernel_avg_ms'], r['roofline'].get('traffic_stale'), r.get('parity_check'))
for k,v in r['config']['others'].items():
    if isinstance(v, dict) and 'ms_per_step' in v: print(k, v['kernel'], round(v['ms_per_step'],4), '%.3g'%v['frames_per_s'], round(v['hbm_frac'],3))
print(r.get('host_to_host')); print(r['cpu_baseline']['value'], r['cpu_baseline'].get('all_cores'))
PY
