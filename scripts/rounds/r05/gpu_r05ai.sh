#!/bin/bash
# round 5, GPU pass ai: A/B on one box -- mel sums / chroma gather as 64 lane jobs (default) against one lane per filter / class
# (-DPAA_TRI_LANE_JOBS=0 build), alternating
out=gpurun_out/r05ai; mkdir -p $out
for i in 1 2; do
for c in w1024 w2048 w512 w2400 w1764 reg_features_stereo w551_11k reg_chromagram_stereo; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "lane_jobs", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_old.so timeout 200 python scripts/kernel_loop.py --case $c --launches 60 | sed 's/^{/{"lib": "per_owner", /' >> $out/loops.jsonl 2>> $out/loops.err
done
done
python - <<'PY'
import json, collections
r = collections.OrderedDict()
for ln in open('gpurun_out/r05ai/loops.jsonl'):
    d = json.loads(ln); r.setdefault(d['case'], {}).setdefault(d['lib'], []).append(d['ms_per_step'])
for c, v in r.items():
    a, b = sum(v['lane_jobs']) / len(v['lane_jobs']), sum(v['per_owner']) / len(v['per_owner'])
    print(c, 'lane jobs', ['%.4f' % x for x in v['lane_jobs']], 'per owner', ['%.4f' % x for x in v['per_owner']], '%+.1f %%' % (100 * (a / b - 1)))
PY
tail -3 $out/loops.err
