#!/bin/bash
# round 6, the round's evidence pass on the build in the tree: whole -m gpu suite, default bench line, headline kernel trace +
# counter passes (scripts/profile.sh), the wave trace of the headline kernel (trace build) and the device-code record.
# usage: bash scripts/rounds/r06/gpu_round_r06.sh <tag>       -> gpurun_out/<tag>/ ; scripts/round_records.py <tag> turns it into profiles/r06_*
tag=${1:-r06h}
out=gpurun_out/$tag; mkdir -p $out
(timeout 1500 python -m pytest tests -m gpu -q --no-header --durations=6 2>&1 | tail -16) > $out/tests.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
cp gpurun_out/bench_full_n1.json $out/bench_full_n1.json 2>/dev/null
timeout 900 bash scripts/profile.sh r06 > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r06 gpurun_out/r06_fast800_w8_summary.json > $out/summarize.log 2>&1
rm -rf gpurun_out/prof_r06/trace gpurun_out/prof_r06/pmc1 gpurun_out/prof_r06/pmc2 gpurun_out/prof_r06/pmc3 gpurun_out/prof_r06/pmc4
if [ -f pyaudioanalysis_amd/libpaa_hip_trace.so ]; then
(echo "# scripts/phase_timing.py with the -DPAA_F800_TRACE build (per-wave life times only), headline plan (1-hour clip, 800/400), after 3 s of untimed launches"
 PAA_PHASE_PREWARM=3 PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trace.so timeout 300 python scripts/phase_timing.py | grep -v "0.00 %") > $out/wave_trace.txt 2> $out/wave_trace.err
fi
python scripts/device_code_hash.py --write $out/device_code.json --note "pytest -m gpu + bench + profiles of gpurun_out/$tag" > /dev/null 2>&1
tail -6 $out/tests.log
python - <<PY
import json
s = open('$out/bench.json').read().strip().splitlines()[-1]
d = json.loads(s)
print('line chars', len(s), 'value', d['value'], 'ms', d['ms_per_step'], 'kernel ms', d['roofline']['kernel_avg_ms'], d.get('parity_check', {}).get('status'))
for k, v in d['configs'].items(): print(' ', k, v)
PY
