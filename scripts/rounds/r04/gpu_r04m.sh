#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
(timeout 600 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -6) > $out/tests.log
timeout 300 python scripts/host_breakdown.py > $out/host.log 2>&1
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
head -3 $out/tests.log; cat $out/host.log; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04m/bench.json').read().strip().split('\n')[-1])
print(r['value'], r['ms_per_step'], r['roofline']['kernel_avg_ms'])
print(r.get('host_to_host'))
PY
