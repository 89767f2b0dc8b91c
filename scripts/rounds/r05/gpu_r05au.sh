#!/bin/bash
# round 5, GPU pass au: the default bench line and the counter passes of 2048/1024 on the final build (pass-3 fold of 16 x 16 x 4; every
# other translation unit's device code is the one of pass al -- profiles/r05_device_code.json)
out=gpurun_out/r05au; mkdir -p $out
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05au/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
for k in ('cfg2', 'cfg5_features', 'w1024_16kHz', 'w2048_44kHz', 'w512_16kHz', 'w551_11kHz', 'w16000_16kHz', 'w44100_44kHz'): print(k, d['configs'][k])
PY
timeout 200 bash scripts/profile_kernel.sh r05 w2048 40 > $out/prof_w2048.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r05_w2048_summary.json')); print(d.get('kernel_avg_us'), d['run_under_trace']['ms_per_step'], d.get('lds_bank_conflict_ratio'), d.get('valu_issue_fraction'), d['traffic']['traffic_over_algorithmic'])"
