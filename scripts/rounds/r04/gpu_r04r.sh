#!/bin/bash
out=gpurun_out/r04r; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_ct_kernels_gpu.py -q --no-header -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8) > $out/tests.log
for c in w2400 w2205 reg_features_stereo w551_11k w1764 w2400_68 w2205_stereo_68; do
    timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-150
done > $out/loop.txt
cat $out/tests.log; cat $out/loop.txt
