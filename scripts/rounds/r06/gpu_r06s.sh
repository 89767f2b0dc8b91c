#!/bin/bash
# round 6, GPU pass s: the fused three-pass kernel (kernels_wgr.hpp) -- parity tests of the big windows, loops of the cases, phase split
export TMPDIR=/tmp
out=gpurun_out/r06s; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "big or workgroup or thumbnail" 2>&1 | tail -30) > $out/tests.log
for c in big_16000 big_16000_1h big_16000_68 big_8000_batch; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 50 --warmup 5 2>&1 | tail -1
done > $out/loops.txt 2>&1
PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so python scripts/phase_timing_wgr.py big_16000_1h > $out/phase.txt 2>&1
cat $out/tests.log | tail -15; cat $out/loops.txt | cut -c1-200; cat $out/phase.txt
