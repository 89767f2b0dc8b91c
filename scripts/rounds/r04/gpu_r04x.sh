#!/bin/bash
out=gpurun_out/r04x; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py tests/test_ct_kernels_gpu.py tests/test_parity_at_scale_gpu.py -q --no-header -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Error|assert" | tail -8) > $out/tests.log
for lib in libpaa_hip.so libpaa_hip_ab.so libpaa_hip.so libpaa_hip_ab.so; do
  echo -n "$lib "; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 100 --check 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
  for c in w2400 w2205 ct_640 reg_features_stereo; do
    echo -n "$lib " ; PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib timeout 300 python scripts/kernel_loop.py --case $c --launches 50 2>&1 | tail -1 | cut -c1-140
  done
done > $out/ab.txt 2>&1
cat $out/tests.log; cat $out/ab.txt
