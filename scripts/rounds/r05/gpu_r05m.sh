#!/bin/bash
# round 5, GPU pass m: the split radix-29 butterfly plane by plane (half the live registers: 1102's feature instance 143-159
# instead of 189-207 -> twelve waves per CU), waves per workgroup chosen at plan time (1024: twelve where the tables leave room)
out=gpurun_out/r05m; mkdir -p $out
(timeout 1200 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --maxfail=40 2>&1 | tail -40) > $out/tests.log
tail -6 $out/tests.log
for c in reg_features_stereo reg_features w551_11k w551_22k w1024 w1024_68 w512 w2048 w2400 w2205; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05m/loops.jsonl'):
    d = json.loads(ln); print(d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
