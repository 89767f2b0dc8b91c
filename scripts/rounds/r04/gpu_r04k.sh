#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
for lib in libpaa_hip.so libpaa_hip_ab01.so libpaa_hip_ab10.so libpaa_hip_ab00.so; do
  for c in w2400 reg_features w551_11k w2205; do
    echo -n "$lib " ; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-140
  done
done > $out/ab.txt
cat $out/ab.txt
