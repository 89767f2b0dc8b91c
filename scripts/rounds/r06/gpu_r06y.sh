#!/bin/bash
# round 6, GPU pass y: phase ablation of kernels_wgs.hpp (timing builds -DPAA_WGS_ABLATE=n, results wrong by construction): step time of the
# 44100/22050 bench shape with one phase skipped at a time
export TMPDIR=/tmp
out=gpurun_out/r06y; mkdir -p $out
for a in "" $@; do
  lib=pyaudioanalysis_amd/libpaa_hip${a:+_ab$a}.so
  [ -f $lib ] || continue
  PAA_HIP_LIBRARY=$PWD/$lib timeout 120 python scripts/kernel_loop.py --case big_44100 --launches 50 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ablate %-4s %.4f ms/step' % ('${a:-none}', d['ms_per_step']))"
done 2>&1 | tee $out/ablate.txt
