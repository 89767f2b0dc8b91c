#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r06ab; mkdir -p $out
timeout 600 python -m pytest --timeout 300 tests/test_wgs_kernel_gpu.py -x -q -m gpu > $out/tests.log 2>&1
tail -4 $out/tests.log
for c in big_44100 big_22050 big_44100_20min; do timeout 200 python scripts/kernel_loop.py --case $c --launches 40 --warmup 5; done 2>&1 | cut -c1-190


