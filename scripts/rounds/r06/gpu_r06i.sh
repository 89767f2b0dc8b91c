#!/bin/bash
# round 6, GPU pass i: A/B of the build without the SI load/store optimizer (ds_read2_b64 / ds_write2_b64 cost twice the LDS cycles of
# two single accesses on this chip) on the LDS-heavy families; + the updated ranged-host-call test
out=gpurun_out/r06i; mkdir -p $out
(timeout 600 python -m pytest tests/test_parity_at_scale_gpu.py -m gpu -q --no-header -k ranged 2>&1 | tail -3) | tee $out/tests.log
for lib in libpaa_hip_nolso.so libpaa_hip_nolsv.so libpaa_hip.so; do
  echo "== $lib"
  for c in w2048 w1920 w1024 w512 mix_256 w2400 w2205 w551_11k reg_features_stereo reg_spectrogram_stereo ct_640 ct_800_f64 blu_1103 blu_661 mix_4800 big_16000; do
    PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib python scripts/kernel_loop.py --case $c --launches 30 --warmup 10 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-26s %-28s %.4f ms  %.4g frames/s' % (d['case'], d['kernel'], d['ms_per_step'], d['frames_per_s']))"
  done
  PAA_HIP_LIBRARY=pyaudioanalysis_amd/$lib python bench.py --steps 200 --warmup 100 --no-extras --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done 2>&1 | tee $out/ab.log
