#!/bin/bash
out=gpurun_out/r04o; mkdir -p $out
(timeout 600 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py -q --no-header -x 2>&1 | tail -4) > $out/tests.log
for lib in libpaa_hip.so libpaa_hip_nopace.so libpaa_hip.so libpaa_hip_nopace.so; do
  for c in w2400 w2205 reg_features_stereo w551_11k w1764 w2400_68; do
    echo -n "$lib " ; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-140
  done
done > $out/ab.txt
head -2 $out/tests.log; cat $out/ab.txt
