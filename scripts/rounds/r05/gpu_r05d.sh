#!/bin/bash
# round 5, GPU pass d: where the time of the workgroup-per-frame kernels goes -- ablation builds (-DPAA_WG_ABLATE=<bit>: load,
# time domain, passes, magnitudes of the spectrum kernel; staging, sweeps, roll-off, mel + chroma of the feature kernel), the
# same 50-launch loop of the 16 000 / 8 000 case with each
out=gpurun_out/r05d; mkdir -p $out
timeout 120 python scripts/kernel_loop.py --case big_16000 --launches 50 | sed 's/^{/{"ablate": 0, /' >> $out/ablate.jsonl
for m in 1 2 4 8 16 32 64 128; do
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_abl$m.so timeout 120 python scripts/kernel_loop.py --case big_16000 --launches 50 | sed "s/^{/{\"ablate\": $m, /" >> $out/ablate.jsonl 2>> $out/err.log
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05d/ablate.jsonl'):
    d = json.loads(ln); print(d['ablate'], '%.4f ms' % d['ms_per_step'])
PY
