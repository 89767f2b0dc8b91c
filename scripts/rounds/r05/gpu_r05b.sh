#!/bin/bash
# round 5, GPU pass b: the -m gpu suite with the workgroup-per-frame kernels for the big windows (kernels_wg.hpp) and the fixed
# tests of pass a; A/B of the waves per workgroup of the 1024-sample three-pass kernel (8 / 10 / 11: separate builds);
# loops of the new shapes; counter passes of w1024 and of the 16 000-sample window
out=gpurun_out/r05b; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=8 --maxfail=30 2>&1 | tail -150) > $out/tests.log
tail -12 $out/tests.log
for c in w1024 w1024_68 w512 w2048 big_16000 big_16000_1h big_16000_68 big_8000_batch big_44100; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 >> $out/loops.jsonl 2>> $out/loops.err
done
for nw in 8 10; do
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_nw$nw.so timeout 300 python scripts/kernel_loop.py --case w1024 --launches 50 | sed "s/^{/{\"nw\": $nw, /" >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05b/loops.jsonl'):
    try:
        d = json.loads(ln); print(d.get('nw', ''), d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
    except Exception as e:
        print('?', ln[:200])
PY
tail -5 $out/loops.err
timeout 600 bash scripts/profile_kernel.sh r05 w1024 60 > $out/prof_w1024.log 2>&1
timeout 600 bash scripts/profile_kernel.sh r05 big_16000 30 > $out/prof_big_16000.log 2>&1
ls gpurun_out/*summary.json
