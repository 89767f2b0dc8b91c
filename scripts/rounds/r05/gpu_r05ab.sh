#!/bin/bash
# round 5, GPU pass ab: stability of the per-XCD clocks under the hot kernel (scripts/experiments/xcd_clock_stability.py, trace build)
out=gpurun_out/r05ab; mkdir -p $out
PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trace.so timeout 300 python scripts/experiments/xcd_clock_stability.py > $out/xcd_clock.txt 2> $out/err.txt
cat $out/xcd_clock.txt; tail -2 $out/err.txt
