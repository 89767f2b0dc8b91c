#!/bin/bash
# round 5, GPU pass ad: balanced runs for the three-pass family (3072 runs of 19 / 20 frames instead of 3000 of 20 for config 5) and
# the row-parallel delta expansion: parity of the three-pass / delta tests, then A/B loops on one box against the
# -DPAA_BALANCED_RUNS=0 build (libpaa_hip_equalruns.so), alternating
out=gpurun_out/r05ad; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --maxfail=20 2>&1 | tail -15) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3
for i in 1 2; do
for c in reg_features_stereo reg_spectrogram_stereo w1024 w551_11k w2400 w2048; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "balanced", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_equalruns.so timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "equal", /' >> $out/loops.jsonl 2>> $out/loops.err
done
done
python - <<'PY'
import json, collections
r = collections.defaultdict(list)
for ln in open('gpurun_out/r05ad/loops.jsonl'):
    d = json.loads(ln); r[(d['case'], d['lib'])].append(d['ms_per_step'])
for k in sorted(r): print(k, ['%.4f' % v for v in r[k]])
PY
tail -3 $out/loops.err
