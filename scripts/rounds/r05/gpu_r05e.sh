#!/bin/bash
# round 5, GPU pass e: third version of the workgroup-per-frame kernels (padded LDS buffer against the digit-reversed gather,
# twiddle powers, sign codes once per sample, own row only in the feature kernel's LDS): tests of the path, loops, kernel trace
out=gpurun_out/r05e; mkdir -p $out
(timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q --no-header --maxfail=30 -k "big or workgroup or golden" 2>&1 | tail -40) > $out/tests.log
tail -4 $out/tests.log
for c in big_16000 big_16000_1h big_16000_68 big_8000_batch; do
  timeout 300 python scripts/kernel_loop.py --case $c --launches 50 >> $out/loops.jsonl 2>> $out/loops.err
done
python - <<'PY'
import json
for ln in open('gpurun_out/r05e/loops.jsonl'):
    d = json.loads(ln); print(d['case'], d['kernel'], '%.4f ms' % d['ms_per_step'], '%.3g frames/s' % d['frames_per_s'])
PY
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python scripts/kernel_loop.py --case big_16000 --launches 30 > $out/trace.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r05e/trace/**/*kernel_stats.csv', recursive=True):
    for row in list(csv.DictReader(open(f)))[:4]: print(row['Name'][:70], row['Calls'], row['AverageNs'])
PY
rm -rf $out/trace
