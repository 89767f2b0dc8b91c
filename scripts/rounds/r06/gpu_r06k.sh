#!/bin/bash
# round 6, GPU pass k: the Bluestein kernel with the four-pass length 8192 (windows 2732 .. 5461)
out=gpurun_out/r06k; mkdir -p $out
(timeout 900 python -m pytest tests/test_blu_kernel_gpu.py -m gpu -q --no-header 2>&1 | tail -8 | cut -c1-400) | tee $out/tests.log
for c in blu_736 blu_4001 blu_2203 blu_1103 blu_661; do python scripts/kernel_loop.py --case $c --launches 10 --warmup 3; done 2>&1 | cut -c1-200 | tee $out/loop.log
