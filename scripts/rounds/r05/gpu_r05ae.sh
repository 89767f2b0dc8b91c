#!/bin/bash
# round 5, GPU pass ae: robustness sweep (scripts/random_sweep.py): 600 random (fs, window, step, sample type, deltas) shapes incl. big
# windows, whole matrices against the NumPy oracle with the gates of the parity tests
out=gpurun_out/r05ae; mkdir -p $out
timeout 1500 python scripts/random_sweep.py 600 0 > $out/sweep.txt 2> $out/sweep.err
tail -30 $out/sweep.txt | cut -c1-400; tail -3 $out/sweep.err
