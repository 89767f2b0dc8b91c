#!/bin/bash
# round 5, GPU pass w: A/B of the hot kernel's run geometry on ONE box -- default build (balanced_runs: 2048 runs of 72 / 68 frames
# for the one-hour clip) against the -DPAA_BALANCED_RUNS=0 build (2000 equal runs of 72), alternating, feature kernel only
# (profiling events) and the whole step (kernel_loop)
out=gpurun_out/r05w; mkdir -p $out
for i in 1 2 3; do
  timeout 300 python scripts/experiments/run_geometry.py 143999 143999 | sed 's/^/{"lib": "balanced", "r": /; s/$/}/' >> $out/ab.jsonl
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_equalruns.so timeout 300 python scripts/experiments/run_geometry.py 143999 143999 | sed 's/^/{"lib": "equal", "r": /; s/$/}/' >> $out/ab.jsonl
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05w/ab.jsonl'):
    d = json.loads(ln); print(d['lib'], ['%.4f' % r['kernel_ms'] for r in d['r']])
PY
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('balanced bench', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_equalruns.so timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('equal bench', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
