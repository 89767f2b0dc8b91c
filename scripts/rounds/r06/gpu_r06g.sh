#!/bin/bash
# round 6, GPU pass g: window 256 on the three-pass register FFT (4 x 4 x 8): tests, timing against st_mix's 1.42e8 frames/s, counters
out=gpurun_out/r06g; mkdir -p $out
(timeout 900 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py -m gpu -q --no-header 2>&1 | tail -12 | cut -c1-600) | tee $out/tests.log
python scripts/kernel_loop.py --case mix_256 --launches 20 --warmup 5 2>&1 | cut -c1-230 | tee $out/loop.log
timeout 400 bash scripts/profile_kernel.sh r06 mix_256 20 > $out/prof_256.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r06_mix_256_summary.json')); print(d['kernel_avg_us'], d['traffic']['traffic_over_algorithmic'], d['lds_bank_conflict_ratio'], d['valu_issue_fraction'], d['run_under_trace']['frames_per_s'])"
