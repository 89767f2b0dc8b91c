#!/bin/bash
# round 6, GPU pass b: first run of the Bluestein kernel (kernels_blu.hpp) -- its own test file, the goldens of the new windows
out=gpurun_out/r06b; mkdir -p $out
(timeout 1200 python -m pytest tests/test_blu_kernel_gpu.py -m gpu -q --no-header 2>&1 | grep -E "^(FAILED|E  +Assertion|[0-9]+ (passed|failed))" | cut -c1-900 | tail -60) > $out/tests_blu.log
(timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py -m gpu -q --no-header 2>&1 | grep -E "^(FAILED|E  +Assertion|[0-9]+ (passed|failed))" | cut -c1-900 | tail -60) > $out/tests_parity.log
cat $out/tests_blu.log; tail -15 $out/tests_parity.log
