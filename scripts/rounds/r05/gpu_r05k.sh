#!/bin/bash
# round 5, GPU pass k (second run of pass j, rows instances at twelve waves per workgroup): config 5's spectrogram / chromagram rows on the prime-factor kernel (st_reg, default) against the
# three-pass kernel with the split radix-29 pass (-DPAA_TRI_1102_ROWS=1 build)
out=gpurun_out/r05k; mkdir -p $out
for c in reg_spectrogram_stereo reg_chromagram_stereo reg_spectrogram; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 100 | sed 's/^{/{"lib": "default", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trirows.so timeout 300 python scripts/kernel_loop.py --case $c --launches 100 | sed 's/^{/{"lib": "tri_rows", /' >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05k/loops.jsonl'):
    d = json.loads(ln); print(d['lib'], d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
