#!/bin/bash
# round 5, GPU pass c: second version of the workgroup-per-frame kernels (two-level LDS twiddles, two butterflies per lane, rows
# staged in LDS by the feature kernel) and the group pitch of the radix-8 shapes' second exchange: tests of the touched paths,
# loops, counter passes
out=gpurun_out/r05c; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_mix_kernel_gpu.py tests/test_similarity_gpu.py -m gpu -q --no-header --maxfail=30 2>&1 | tail -60) > $out/tests.log
tail -6 $out/tests.log
for c in w1024 w1024_68 w1024_spectrogram w512 w2048 big_16000 big_16000_1h big_16000_68 big_8000_batch; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05c/loops.jsonl'):
    try:
        d = json.loads(ln); print(d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
    except Exception as e:
        print('?', ln[:200])
PY
tail -5 $out/loops.err
timeout 600 bash scripts/profile_kernel.sh r05 w1024 60 > $out/prof_w1024.log 2>&1
timeout 600 bash scripts/profile_kernel.sh r05 big_16000 30 > $out/prof_big_16000.log 2>&1
python - <<'PY'
import json
for c in ('w1024', 'big_16000'):
    d = json.load(open('gpurun_out/r05_%s_summary.json' % c))
    print(c, [(k['name'][:60], k['avg_us']) for k in d['kernel_trace_stats'][:3]], d.get('lds_bank_conflict_ratio'), d.get('traffic', {}).get('traffic_over_algorithmic'))
PY
