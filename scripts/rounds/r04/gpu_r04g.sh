#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
(timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -q --no-header -x 2>&1 | tail -8) > $out/tests.log
for c in reg_features reg_features_stereo reg_spectrogram reg_spectrogram_stereo reg_chromagram_stereo; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1
  PAA_REG_1102=1 PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_exp.so timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1
done > $out/cases.jsonl
tail -5 $out/tests.log; cut -c1-175 $out/cases.jsonl
