#!/bin/bash
# round 6, GPU pass c: first timing of the Bluestein kernel (the in-tree build at call time) + counter passes of 1103 / 661
out=gpurun_out/r06c; mkdir -p $out
for c in blu_1103 blu_661 blu_736 blu_1103_spectrogram; do python scripts/kernel_loop.py --case $c --launches 20 --warmup 5; done > $out/loop.log 2>&1
cat $out/loop.log | cut -c1-260
timeout 300 bash scripts/profile_kernel.sh r06c blu_1103 20 > $out/prof_blu_1103.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r06c_blu_1103_summary.json')); print(json.dumps(d)[:3000])"
