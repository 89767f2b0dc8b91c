#!/bin/bash
# round 5, GPU pass al (seventh consolidation): statistics pass on packed 16-bit operations and chunks of up to 128 K samples: the
# whole -m gpu suite, the default bench line, the headline profile passes (the prologue of the hot kernel folds half the partials),
# counter passes of config 5's rows (stereo statistics with the new chunking)
out=gpurun_out/r05al; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05al/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05al/bench.err').read()[-2000:])
PY
timeout 900 bash scripts/profile.sh r05 > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r05 gpurun_out/r05_fast800_w8_summary.json > $out/summarize.log 2>&1
rm -rf gpurun_out/prof_r05/trace gpurun_out/prof_r05/pmc1 gpurun_out/prof_r05/pmc2 gpurun_out/prof_r05/pmc3 gpurun_out/prof_r05/pmc4
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05_fast800_w8_summary.json'))
for k in d['kernel_trace_stats'][:3]: print(k['name'][:70], k['calls'], k['avg_us'])
print(d.get('traffic'))
PY
for c in reg_spectrogram_stereo reg_features_stereo fast_s800; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
  python - $c <<'PY'
import json, sys
d = json.load(open('gpurun_out/r05_%s_summary.json' % sys.argv[1]))
for k in d['kernel_trace_stats'][:2]: print(sys.argv[1], k['name'][:70], k['calls'], k['avg_us'])
PY
done
