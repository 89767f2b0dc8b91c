#!/bin/bash
# round 5, GPU pass i: the prime butterflies of the 551 / 1102 shapes shared by three lanes (SplitSel): tests of the three-pass
# family and of config 5 at scale, loops
out=gpurun_out/r05i; mkdir -p $out
(timeout 1200 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --maxfail=40 2>&1 | tail -80) > $out/tests.log
tail -30 $out/tests.log
for c in reg_features_stereo reg_features w551_11k w551_22k w2205; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05i/loops.jsonl'):
    d = json.loads(ln); print(d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
