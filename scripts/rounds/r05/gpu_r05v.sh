#!/bin/bash
# round 5, GPU pass v: the hot kernel's runs re-cut to fill all 256 CUs (lib_plan.hpp: balanced_runs; 1184 runs of 72 + 864 of 68
# frames instead of 2000 of 72): parity tests of the fast kernel, the geometry experiment, the default bench line
out=gpurun_out/r05v; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --maxfail=20 2>&1 | tail -15) > $out/tests.log
tail -3 $out/tests.log
timeout 600 python scripts/experiments/run_geometry.py > $out/run_geometry.json 2> $out/err.log
cat $out/run_geometry.json; tail -3 $out/err.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05v/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
for k in ('cfg2', 'cfg3', 'cfg4_shard', 'step800_68rows'): print(k, d['configs'][k])
PY
