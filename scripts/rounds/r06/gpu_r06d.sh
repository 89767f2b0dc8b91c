#!/bin/bash
# round 6, GPU pass d: Bluestein kernel after the staging tile left the LDS (four waves per CU at M = 2048): timing + per-phase cycles
out=gpurun_out/r06d; mkdir -p $out
for c in blu_1103 blu_661 blu_736 blu_1103_spectrogram; do python scripts/kernel_loop.py --case $c --launches 20 --warmup 5; done 2>&1 | cut -c1-200 | tee $out/loop.log
PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so python scripts/phase_timing_blu.py blu_1103 blu_661 2>&1 | tee $out/phases.log
(timeout 600 python -m pytest tests/test_blu_kernel_gpu.py -m gpu -q --no-header 2>&1 | tail -5) | tee $out/tests_blu.log
