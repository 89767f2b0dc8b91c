#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so timeout 300 python scripts/phase_timing_tri.py w2400 w2205 > $out/phases.log 2>&1
for c in w2400 w2205; do timeout 900 bash scripts/profile_kernel.sh r04 $c > $out/prof_$c.log 2>&1; done
cat $out/phases.log
