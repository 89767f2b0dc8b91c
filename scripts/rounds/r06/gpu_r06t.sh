#!/bin/bash
# round 6, GPU pass t: the fused three-pass kernel -- its new tests + the big-window tests, loops, kernel trace + counter passes
export TMPDIR=/tmp
out=gpurun_out/r06t; mkdir -p $out
(timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header -x -k "big or workgroup or thumbnail or fused" --durations=5 2>&1 | tail -30) > $out/tests.log
for c in big_16000 big_16000_1h big_16000_68 big_8000_batch; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 50 --warmup 5 2>&1 | tail -1
done > $out/loops.txt 2>&1
for c in big_16000 big_16000_1h; do
  timeout 300 bash scripts/profile_kernel.sh r06 $c 20 > $out/prof_$c.log 2>&1
  python -c "
import json
d=json.load(open('gpurun_out/r06_${c}_summary.json')); print('%-28s %-30s %8.1f us  traffic %s  conflicts %s  issue %s' % ('$c', d['run_under_trace']['kernel'], d.get('kernel_avg_us') or 0, d.get('traffic',{}).get('traffic_over_algorithmic'), d.get('lds_bank_conflict_ratio'), d.get('valu_issue_fraction')))"
done 2>&1 | tee $out/summary.txt
cat $out/tests.log | tail -15; cat $out/loops.txt | cut -c1-200
