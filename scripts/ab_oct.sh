#!/bin/bash
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  PAA_HIPCC_FLAGS="$flags" python -c "from pyaudioanalysis_amd import _build; _build.build(force=True)" || continue
  echo "flags=[$flags]"
  PAA_HIP_FAST_OCT=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ', d['config']['kernel'], 'frames/s %.4g  kernel_ms %.4f  %s' % (d['value'], d['roofline']['kernel_avg_ms'], d['parity_spot_check']))"
done
python -c "from pyaudioanalysis_amd import _build; _build.build(force=True)"
