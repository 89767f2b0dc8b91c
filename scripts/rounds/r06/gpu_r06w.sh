#!/bin/bash
# round 6, GPU pass w: kernels_wgs.hpp (44 100- / 22 050-sample windows on the real-input split + three register passes): its tests, the
# big-window tests, the goldens, the loop of the bench shape; optional: kernel trace + counters (argument "prof")
export TMPDIR=/tmp
out=gpurun_out/r06w; mkdir -p $out
timeout 600 python -m pytest --timeout 200 tests/test_wgs_kernel_gpu.py tests/test_parity_gpu.py tests/test_similarity_gpu.py -x -q -m gpu -k "wgs or big_window or workgroup_lds or big_windows or golden or thumb" > $out/tests.log 2>&1
tail -15 $out/tests.log
for c in big_44100; do timeout 120 python scripts/kernel_loop.py --case $c --launches 50; done 2>&1 | tee $out/loop.log | cut -c1-200
if [ "$1" = "prof" ]; then bash scripts/rounds/r06/gpu_r06x.sh r06x big_44100; fi
