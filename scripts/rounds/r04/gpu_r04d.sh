#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py -q --no-header -x 2>&1 | tail -40) > $out/tests.log
for c in w2400 w2205 w2400_68 w2205_stereo_68 w1764 w1920 w551_11k w551_22k; do timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1; done > $out/cases.jsonl
tail -30 $out/tests.log; cat $out/cases.jsonl | cut -c1-200
