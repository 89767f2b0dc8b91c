#!/bin/bash
out=gpurun_out/r04s; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_ct_kernels_gpu.py -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -12) > $out/tests.log
cat $out/tests.log
