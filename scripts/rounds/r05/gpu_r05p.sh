#!/bin/bash
# round 5, GPU pass p: split transforms (wg_split_kernel: windows whose sequence exceeds one CU's LDS, 44 100 samples and up) --
# parity of the big-window tests, the ranged host-call tests with the corrected frame count, the statistics pass with four
# chunks per CU on short batches (cfg5 rows), loops of the affected bench shapes
out=gpurun_out/r05p; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --durations=8 --maxfail=20 -k "big or workgroup or ranged or golden or window" 2>&1 | tail -60) > $out/tests.log
tail -8 $out/tests.log
for c in big_44100 big_16000 reg_spectrogram_stereo reg_features_stereo ct_640_spectrogram; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 30 2>&1 | tail -1 | cut -c1-400
done
