#!/bin/bash
# round 6, GPU pass z: loops of the kernels_wgs.hpp shapes (5 min / 20 min at 44.1 kHz, 20 min at 22.05 kHz)
export TMPDIR=/tmp
out=gpurun_out/r06z; mkdir -p $out
for c in big_44100 big_44100_20min big_22050 big_48000 big_32000; do timeout 200 python scripts/kernel_loop.py --case $c --launches 30 --warmup 5; done 2>&1 | tee $out/loop.log | cut -c1-190
