#!/bin/bash
# round 6, GPU pass sw (final build, after kernels_wgs.hpp): robustness sweep (scripts/random_sweep.py): the 600 random (fs, window, step, sample type, deltas) shapes of round 5
# (same seeds) on this round's binary -- which kernel takes each shape now (188 of 600 were on st_generic)
out=gpurun_out/r06sw; mkdir -p $out
timeout 2400 python scripts/random_sweep.py 600 0 > $out/sweep.txt 2> $out/sweep.err
tail -30 $out/sweep.txt | cut -c1-600; tail -3 $out/sweep.err
