#!/bin/bash
# round 5, GPU pass f: counter passes of the workgroup-per-frame kernels (third version)
out=gpurun_out/r05f; mkdir -p $out
timeout 600 bash scripts/profile_kernel.sh r05 big_16000 30 > $out/prof_big_16000.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_big_16000_summary.json'))
for k in d['kernel_trace_stats'][:4]: print(k['name'][:70], k['calls'], k['avg_us'])
print({k: v for k, v in d.items() if k not in ('kernel_trace_stats', 'pmc', 'pmc_feature_kernel', 'run_under_trace')})
print(json.dumps(d.get('pmc') or d.get('pmc_feature_kernel'), indent=0)[:3000])
PY
