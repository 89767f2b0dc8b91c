#!/bin/bash
# round 5, GPU pass o (third consolidation): the whole -m gpu suite on the build with config 5's rows on the three-pass kernel, the
# plane-wise shared prime butterflies and the 12-wave power-of-two shapes; the default bench line; the counter passes of every
# three-pass case (all of them changed since pass l)
out=gpurun_out/r05o; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05o/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05o/bench.err').read()[-2000:])
PY
for c in w1024 w1024_spectrogram w512 w2048 w2400 w2205 w1764 w1920 w551_11k reg_features_stereo reg_spectrogram_stereo reg_chromagram_stereo; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
ls gpurun_out/r05_*_summary.json | wc -l
