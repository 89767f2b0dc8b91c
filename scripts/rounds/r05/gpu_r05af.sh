#!/bin/bash
# round 5, GPU pass af: per-phase cycle split of the three-pass shapes of round 5 (timing build, scripts/phase_timing_tri.py)
out=gpurun_out/r05af; mkdir -p $out
PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_timing.so timeout 300 python scripts/phase_timing_tri.py w2048 w1024 w512 w2400 reg_features_stereo reg_spectrogram_stereo w551_11k > $out/phases.txt 2> $out/err.txt
cat $out/phases.txt; tail -2 $out/err.txt
