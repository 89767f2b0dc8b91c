#!/bin/bash
# round 5, GPU pass ah: mel sums and chroma gather of the three-pass kernels on all 64 lanes (lane jobs + segmented row scan): parity
# of every three-pass test (features, chromagram rows, goldens), then loops of the feature / chromagram shapes
out=gpurun_out/r05ah; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_ct_kernels_gpu.py -m gpu -q --no-header --maxfail=20 2>&1 | tail -25) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3; grep -n "^FAILED\|^E " $out/tests.log | head -20
for c in w1024 w2048 w512 w2400 w2205 w1764 w1920 reg_features_stereo w551_11k reg_chromagram_stereo; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 40 >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05ah/loops.jsonl'):
    d = json.loads(ln); print(d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.4g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
