#!/bin/bash
# one GPU pass of a round: the whole -m gpu suite, kernel-resident timing of every non-headline case, optional profiles
# usage: bash scripts/gpu_round.sh <tag> [profile-case ...]
tag=${1:-r03}; shift
out=gpurun_out/$tag; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header 2>&1 | tail -60) > $out/tests.log
for c in reg_features reg_features_stereo reg_spectrogram reg_spectrogram_stereo reg_chromagram reg_chromagram_stereo w1024 ct_640 ct_640_spectrogram ct_800_f64 ct_800_stereo ct_400 ct_320 w2400 w2205 mid_stats; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1
done > $out/cases.jsonl
for c in "$@"; do timeout 900 bash scripts/profile_kernel.sh $tag $c > $out/prof_$c.log 2>&1; done
tail -12 $out/tests.log; cat $out/cases.jsonl | cut -c1-330
