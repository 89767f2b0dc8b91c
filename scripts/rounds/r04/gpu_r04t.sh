#!/bin/bash
out=gpurun_out/r04t; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_ct_kernels_gpu.py -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12) > $out/tests.log
for c in ct_640 ct_800_f64 ct_800_stereo ct_400 ct_320 reg_features_stereo w2205; do
    timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-150
done > $out/loop.txt
cat $out/tests.log; cat $out/loop.txt
