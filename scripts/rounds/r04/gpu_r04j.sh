#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
(timeout 600 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -q --no-header -x 2>&1 | tail -6) > $out/tests.log
for c in w2400 w2205 w2400_68 reg_features reg_features_stereo w1764 w1920 w551_11k; do timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1; done > $out/cases.jsonl
head -3 $out/tests.log; cut -c1-170 $out/cases.jsonl
