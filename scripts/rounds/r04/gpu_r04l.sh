#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out
(timeout 600 python -m pytest tests/test_parity_at_scale_gpu.py tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py -q --no-header -x 2>&1 | tail -6) > $out/tests.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
head -3 $out/tests.log; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04l/bench.json').read().strip().split('\n')[-1])
print(r['value'], r['ms_per_step'], r['roofline']['kernel_avg_ms'])
print(r.get('host_to_host'))
for k in ('w2400_48kHz','w2205_44kHz','cfg5_features','w551_11kHz'):
    v=r['config']['others'][k]; print(k, v['kernel'], round(v['ms_per_step'],4), '%.3g'%v['frames_per_s'])
PY
