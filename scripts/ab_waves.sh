set -e
cd $GRAFT_REPO_ROOT
for w in 2 1; do
  PAA_HIPCC_FLAGS="-DPAA_F800_WAVES_PER_SIMD=$w" python -c "from pyaudioanalysis_amd import _build; _build.build(force=True)"
  echo "waves_per_simd=$w"
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_avg_ms'], d['parity_spot_check'])"
done
python -c "from pyaudioanalysis_amd import _build; _build.build(force=True)"
python -m pytest tests -m gpu -q 2>&1 | tail -3
