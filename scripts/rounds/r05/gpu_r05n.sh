#!/bin/bash
# round 5, GPU pass n: spectrogram / chromagram instances of the three-pass family with ONE spectrum slot per wave and up to
# sixteen waves per CU; config 5's rows on st_reg (default) against the three-pass kernel (-DPAA_TRI_1102_ROWS=1 build)
out=gpurun_out/r05n; mkdir -p $out
(timeout 1200 python -m pytest tests/test_mix_kernel_gpu.py tests/test_parity_gpu.py tests/test_parity_at_scale_gpu.py -m gpu -q --no-header --maxfail=40 2>&1 | tail -40) > $out/tests.log
tail -6 $out/tests.log
(PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trirows.so timeout 600 python -m pytest tests/test_parity_at_scale_gpu.py tests/test_parity_gpu.py -m gpu -q --no-header --maxfail=40 -k "config5 or golden or stereo" 2>&1 | tail -8) > $out/tests_trirows.log
tail -3 $out/tests_trirows.log
for c in w1024_spectrogram reg_spectrogram_stereo reg_chromagram_stereo reg_spectrogram; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 100 | sed 's/^{/{"lib": "default", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trirows.so timeout 300 python scripts/kernel_loop.py --case $c --launches 100 | sed 's/^{/{"lib": "tri_rows", /' >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05n/loops.jsonl'):
    d = json.loads(ln); print(d['lib'], d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
