#!/bin/bash
# round 5, GPU pass x: hot kernel with the halo INSIDE a run's first quad (one frame, two with deltas) instead of a halo quad, on
# top of the balanced runs: parity (incl. ranged host call == plan bit for bit), then A/B on one box against the build of pass w
# (libpaa_hip_equalruns.so: halo quad, 2000 equal runs) -- feature kernel by profiling events, and the bench line
out=gpurun_out/r05x; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_directory.py -m gpu -q --no-header --maxfail=20 2>&1 | tail -15) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3
for i in 1 2 3; do
  timeout 300 python scripts/experiments/run_geometry.py 143999 143999 | sed 's/^/{"lib": "halo_inside", "r": /; s/$/}/' >> $out/ab.jsonl
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_equalruns.so timeout 300 python scripts/experiments/run_geometry.py 143999 143999 | sed 's/^/{"lib": "r04_geometry", "r": /; s/$/}/' >> $out/ab.jsonl
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05x/ab.jsonl'):
    d = json.loads(ln); print(d['lib'], ['%.4f' % r['kernel_ms'] for r in d['r']])
PY
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2>$out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05x/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
for k in ('cfg2', 'cfg3', 'cfg4_shard', 'step800_68rows'): print(k, d['configs'][k])
PY
PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_equalruns.so timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r04 geometry bench', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms']); print([ (k, d['configs'][k][:2]) for k in ('cfg3','cfg4_shard','step800_68rows')])"
