#!/bin/bash
out=gpurun_out/r04q; mkdir -p $out
PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so timeout 300 python scripts/phase_timing_ct.py ct_640_spectrogram ct_640_chromagram ct_640 > $out/phases.txt 2>&1
cat $out/phases.txt
