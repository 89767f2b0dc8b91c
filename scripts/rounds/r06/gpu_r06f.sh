#!/bin/bash
# round 6, GPU pass f: counter passes of the Bluestein kernel (1103 / 441 and 661 / 220, one-hour clips) + the fixed tail test
out=gpurun_out/r06f; mkdir -p $out
(timeout 600 python -m pytest tests/test_blu_kernel_gpu.py -m gpu -q --no-header 2>&1 | tail -4 | cut -c1-300) | tee $out/tests.log
timeout 400 bash scripts/profile_kernel.sh r06 blu_1103 20 > $out/prof_blu_1103.log 2>&1
timeout 400 bash scripts/profile_kernel.sh r06 blu_661 20 > $out/prof_blu_661.log 2>&1
python -c "
import json
for c in ('blu_1103','blu_661'):
    d=json.load(open('gpurun_out/r06_%s_summary.json'%c)); d.pop('run_under_trace',None); d.pop('kernel_trace_stats',None); print(c, json.dumps(d)[:2500])"
