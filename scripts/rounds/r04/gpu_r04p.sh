#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py tests/test_parity_at_scale_gpu.py -q --no-header -x -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6) > $out/tests.log
for lib in libpaa_hip.so libpaa_hip_ab.so libpaa_hip.so libpaa_hip_ab.so; do
  for c in ct_640_spectrogram ct_640_chromagram; do
    echo -n "$lib " ; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-170
  done
done > $out/ab.txt
cat $out/tests.log; cat $out/ab.txt
