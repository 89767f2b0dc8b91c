#!/bin/bash
# round 5, GPU pass h (consolidation): the whole -m gpu suite, the default bench line as the driver runs it, and the kernel trace +
# counter passes of every kernel the bench line names (scripts/profile_kernel.sh: trace, issue counters x2, FETCH_SIZE, WRITE_SIZE)
out=gpurun_out/r05h; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
tail -5 $out/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05h/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05h/bench.err').read()[-2000:])
PY
for c in reg_features_stereo reg_spectrogram_stereo reg_chromagram_stereo ct_640 ct_640_spectrogram ct_640_chromagram ct_800_f64 ct_800_stereo ct_400 \
         w1024 w1024_spectrogram w2048 w512 w2400 w2205 w1764 w1920 w551_11k mid_stats fast_s800 big_16000 big_44100 mix_4800 mix_256 generic_1103; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
timeout 300 bash scripts/profile_similarity.sh r05_similarity > $out/prof_sim.log 2>&1
timeout 300 bash scripts/profile_similarity_pmc.sh r05_similarity_pmc > $out/prof_sim_pmc.log 2>&1
rm -rf gpurun_out/prof_r05_similarity/trace gpurun_out/prof_r05_similarity_pmc/p1 gpurun_out/prof_r05_similarity_pmc/p2 gpurun_out/prof_r05_similarity_pmc/p3
ls gpurun_out/r05_*_summary.json | wc -l
