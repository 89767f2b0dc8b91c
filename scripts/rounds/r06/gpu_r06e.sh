#!/bin/bash
# round 6, GPU pass e: Bluestein kernel -- tests of the family, timing on one-hour clips, kernel trace + counter passes of 1103 / 661
out=gpurun_out/r06e; mkdir -p $out
(timeout 900 python -m pytest tests/test_blu_kernel_gpu.py tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py -m gpu -q --no-header 2>&1 | tail -25 | cut -c1-400) | tee $out/tests.log
for c in blu_1103 blu_661 blu_736 blu_1103_spectrogram blu_2203 blu_202; do python scripts/kernel_loop.py --case $c --launches 20 --warmup 5; done 2>&1 | cut -c1-200 | tee $out/loop.log
timeout 400 bash scripts/profile_kernel.sh r06 blu_1103 20 > $out/prof_blu_1103.log 2>&1
timeout 400 bash scripts/profile_kernel.sh r06 blu_661 20 > $out/prof_blu_661.log 2>&1
python -c "
import json
for c in ('blu_1103','blu_661'):
    d=json.load(open('gpurun_out/r06_%s_summary.json'%c)); print(c, json.dumps(d)[:1800])"
