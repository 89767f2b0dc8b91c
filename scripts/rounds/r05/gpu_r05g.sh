#!/bin/bash
# round 5, GPU pass g: persistent spectrum kernel (tables once per workgroup, next frame's records and samples fetched under the
# magnitude pass) at 512 / 768 / 1024 threads per workgroup (separate builds), tests of the path on the default build
out=gpurun_out/r05g; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header --maxfail=30 -k "big or workgroup or golden" 2>&1 | tail -40) > $out/tests.log
tail -4 $out/tests.log
for c in big_16000 big_16000_1h big_8000_batch; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 | sed 's/^{/{"nt": 0, /' >> $out/loops.jsonl 2>> $out/loops.err
  for nt in 512 768; do
    PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_nt$nt.so timeout 300 python scripts/kernel_loop.py --case $c --launches 50 | sed "s/^{/{\"nt\": $nt, /" >> $out/loops.jsonl 2>> $out/loops.err
  done
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05g/loops.jsonl'):
    d = json.loads(ln); print(d['nt'], d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
tail -3 $out/loops.err
