#!/bin/bash
# round 6, GPU pass a: the whole -m gpu suite on the tree after st_reg left it, the default bench line through the new formatter
# (must be < 8 000 characters and parse), and the 20-run cold-start stress of two processes on one device
out=gpurun_out/r06a; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=8 -x 2>&1 | tail -20) > $out/tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
s = open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1]
d = json.loads(s)
print('line chars', len(s), 'value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
print('cpu_baseline', d['cpu_baseline'])
PY
timeout 1500 python scripts/two_proc_stress.py --runs 20 --out $out/two_proc_stress.txt > $out/stress.log 2>&1
tail -5 $out/stress.log
tail -12 $out/tests.log
