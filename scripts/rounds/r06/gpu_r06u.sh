#!/bin/bash
# round 6, GPU pass u: 16-byte pass-3 reads of the three-pass kernel (R3 = 2 / 4): tests of the family, loops + profiles of 2048 / 1764
export TMPDIR=/tmp
out=gpurun_out/r06u; mkdir -p $out
(timeout 1200 python -m pytest tests/test_tri_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q --no-header -x -k "tri or pow2 or 2048 or 1764 or 1600 or golden" 2>&1 | tail -8) > $out/tests.log
for c in w2048 w1764 w1920 w1024; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 100 --warmup 10 2>&1 | tail -1 | cut -c1-200
done > $out/loops.txt 2>&1
for c in w2048 w1764; do
  timeout 300 bash scripts/profile_kernel.sh r06 $c 20 > $out/prof_$c.log 2>&1
  python -c "
import json
d=json.load(open('gpurun_out/r06_${c}_summary.json')); print('%-28s %-30s %8.1f us  traffic %s  conflicts %s  issue %s' % ('$c', d['run_under_trace']['kernel'], d.get('kernel_avg_us') or 0, d.get('traffic',{}).get('traffic_over_algorithmic'), d.get('lds_bank_conflict_ratio'), d.get('valu_issue_fraction')))"
done 2>&1 | tee $out/summary.txt
cat $out/tests.log; cat $out/loops.txt
