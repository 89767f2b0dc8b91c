#!/bin/bash
# round 5, GPU pass ak (sixth consolidation): the whole -m gpu suite on the build with the lane jobs of the three-pass family, the
# balanced runs of the 2 RA RB family and the row-parallel delta expansion; the default bench line; counter passes of every
# three-pass and 2 RA RB case (all changed since passes o / h); the kernel trace of the small kernels
out=gpurun_out/r05ak; mkdir -p $out
(timeout 1200 python -m pytest tests -m gpu -q --no-header --durations=5 --maxfail=30 2>&1 | tail -60) > $out/tests.log
grep -n "passed\|failed" $out/tests.log | tail -3
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r05ak/bench.json').read().strip().splitlines()[-1])
    print('value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
    for k, v in d['configs'].items(): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/r05ak/bench.err').read()[-2000:])
PY
for c in w1024 w1024_spectrogram w512 w2048 w2400 w2205 w1764 w1920 w551_11k reg_features_stereo reg_spectrogram_stereo reg_chromagram_stereo ct_400 ct_640 ct_640_spectrogram ct_640_chromagram ct_800_f64 ct_800_stereo; do
  timeout 300 bash scripts/profile_kernel.sh r05 $c 40 > $out/prof_$c.log 2>&1
done
export TMPDIR=/tmp
aux=$GRAFT_REPO_ROOT/gpurun_out/prof_r05_aux_kernels; mkdir -p $aux
rocprofv3 --kernel-trace --stats -d $aux/trace -o trace -- python scripts/aux_kernels_loop.py > $aux/run.log 2>&1
python scripts/summarize_aux_prof.py $aux gpurun_out/r05_aux_kernels_summary.json | grep -i "expand\|beat" 
rm -rf $aux/trace
ls gpurun_out/r05_*_summary.json | wc -l
