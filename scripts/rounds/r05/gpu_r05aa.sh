#!/bin/bash
# round 5, GPU pass aa: where the +-5 % spread of the hot kernel's workgroup lives comes from (per XCD / per workgroup), wave trace build
out=gpurun_out/r05aa; mkdir -p $out
PAA_PHASE_PREWARM=2 PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trace.so timeout 300 python scripts/phase_timing.py 2> $out/err.txt | grep -v "0.00 %" > $out/wave_trace.txt
tail -16 $out/wave_trace.txt; tail -2 $out/err.txt
