#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
(timeout 420 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -30) > $out/tests.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
tail -12 $out/tests.log; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04e/bench.json').read().strip().split('\n')[-1])
print(r['value'], r['ms_per_step'], r['roofline']['kernel_avg_ms'], r.get('parity_check'))
for k,v in r['config']['others'].items():
    if isinstance(v, dict) and 'ms_per_step' in v: print(k, v['kernel'], round(v['ms_per_step'],4), '%.3g'%v['frames_per_s'], round(v['hbm_frac'],3))
print(r.get('host_to_host'))
PY
