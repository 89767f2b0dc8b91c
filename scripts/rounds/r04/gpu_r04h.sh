#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
(timeout 600 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -6) > $out/tests.log
for c in reg_features_stereo reg_spectrogram_stereo w551_11k; do timeout 600 bash scripts/profile_kernel.sh r04 $c > $out/prof_$c.log 2>&1; done
for c in reg_features reg_features_stereo reg_spectrogram reg_spectrogram_stereo reg_chromagram_stereo w1024 w551_11k mid_stats; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1
done > $out/cases.jsonl
head -4 $out/tests.log; cut -c1-175 $out/cases.jsonl
