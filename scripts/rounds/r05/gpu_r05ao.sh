#!/bin/bash
# round 5, GPU pass ao: A/B -- two radix-8 butterflies in flight per lane in the 768-thread spectrum kernel (-DPAA_WG_U8=1 build, 153
# registers) against one (125 registers), alternating
out=gpurun_out/r05ao; mkdir -p $out
for i in 1 2; do
for c in big_16000 big_16000_1h big_8000_batch; do
  timeout 200 python scripts/kernel_loop.py --case $c --launches 40 | sed 's/^{/{"lib": "u1", /' >> $out/loops.jsonl 2>> $out/loops.err
  PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_u8.so timeout 200 python scripts/kernel_loop.py --case $c --launches 40 | sed 's/^{/{"lib": "u2", /' >> $out/loops.jsonl 2>> $out/loops.err
done
done
python - <<'PY'
import json, collections
r = collections.OrderedDict()
for ln in open('gpurun_out/r05ao/loops.jsonl'):
    d = json.loads(ln); r.setdefault(d['case'], {}).setdefault(d['lib'], []).append(d['ms_per_step'])
for c, v in r.items():
    a, b = sum(v['u2']) / len(v['u2']), sum(v['u1']) / len(v['u1'])
    print(c, 'u2', ['%.4f' % x for x in v['u2']], 'u1', ['%.4f' % x for x in v['u1']], '%+.1f %%' % (100 * (a / b - 1)))
PY
tail -3 $out/loops.err
