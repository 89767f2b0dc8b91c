#!/bin/bash
# round 5, GPU pass ag: what the mel sums (40 of 64 lanes, the widest filter sets the trip count) and the chroma gather (12 of 64
# lanes) cost the three-pass feature kernels: builds with -DPAA_TRI_ABLATE=1 / 2 / 3 (no mel, no chroma, neither) against the product
out=gpurun_out/r05ag; mkdir -p $out
for c in w1024 w2048 w512 w2400 reg_features_stereo w551_11k; do
  for lib in default abl1 abl2 abl3; do
    if [ $lib = default ]; then L=$PWD/pyaudioanalysis_amd/libpaa_hip.so; else L=$PWD/pyaudioanalysis_amd/libpaa_hip_$lib.so; fi
    PAA_HIP_LIBRARY=$L timeout 200 python scripts/kernel_loop.py --case $c --launches 40 | sed "s/^{/{\"lib\": \"$lib\", /" >> $out/loops.jsonl 2>> $out/loops.err
  done
done
python - <<'PY'
import json, collections
r = collections.OrderedDict()
for ln in open('gpurun_out/r05ag/loops.jsonl'):
    d = json.loads(ln); r.setdefault(d['case'], {})[d['lib']] = d['ms_per_step']
for c, v in r.items(): print(c, {k: '%.4f' % x for k, x in v.items()}, 'mel %.1f %% chroma %.1f %% both %.1f %%' % tuple(100 * (1 - v[k] / v['default']) for k in ('abl1', 'abl2', 'abl3')))
PY
tail -3 $out/loops.err
