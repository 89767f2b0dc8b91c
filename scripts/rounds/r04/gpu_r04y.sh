#!/bin/bash
# final pass of round 4: gate on the 1102 row kernels (changed last), A/B of their timing, then the round's evidence
out=gpurun_out/r04y; mkdir -p $out
(timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py tests/test_mix_kernel_gpu.py -q --no-header -p no:cacheprovider -k "spec or chroma or 1102 or config5 or cfg5 or dispatch" 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8) > $out/gate.log
cat $out/gate.log
if grep -q "failed\|Error" $out/gate.log; then echo GATE FAILED; exit 1; fi
for lib in libpaa_hip.so libpaa_hip_ab.so libpaa_hip.so libpaa_hip_ab.so; do
  for c in reg_spectrogram_stereo reg_chromagram_stereo; do
    echo -n "$lib " ; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-150
  done
done > $out/ab.txt 2>&1
cat $out/ab.txt
bash scripts/gpu_round.sh r04 w2400 w2205 reg_features_stereo reg_spectrogram_stereo
