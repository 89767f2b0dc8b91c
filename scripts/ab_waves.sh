mkdir -p gpurun_out/r02b
for w in 8 4; do
  PAA_F800_WAVES=$w python -m pytest tests/test_parity_gpu.py -q -x -k "short_term_golden or seeded or known or batch_equals or size_independent" > gpurun_out/r02b/pytest_w$w.log 2>&1; tail -3 gpurun_out/r02b/pytest_w$w.log
  PAA_F800_WAVES=$w python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02b/bench_w$w.json 2> gpurun_out/r02b/bench_w$w.err; python -c "
import json; d=json.load(open('gpurun_out/r02b/bench_w$w.json')); print('waves $w', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['config']['kernel'], d.get('parity_spot_check'))"
done
