#!/bin/bash
# round 5, GPU pass t (fourth consolidation): the whole -m gpu suite on the build with the split transforms (wg_split_kernel), the
# default bench line, the counter passes of the workgroup-per-frame kernels (the only ones that changed since pass o)
out=gpurun_out/r05t; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05t/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05t/bench.err').read()[-2000:])
PY
for c in big_16000 big_44100; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
ls gpurun_out/r05_*_summary.json | wc -l
