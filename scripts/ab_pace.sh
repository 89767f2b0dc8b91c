mkdir -p gpurun_out/r02c
for pm in 0 1 2; do
  PAA_F800_PACE=$pm python -m pytest tests/test_parity_gpu.py -q -x -k "short_term_golden or seeded or batch_equals" > gpurun_out/r02c/pytest_p$pm.log 2>&1; tail -1 gpurun_out/r02c/pytest_p$pm.log
  for rep in 1 2; do
  PAA_F800_PACE=$pm python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02c/bench_p$pm.json 2> gpurun_out/r02c/bench_p$pm.err; python -c "
import json; d=json.load(open('gpurun_out/r02c/bench_p$pm.json')); print('pace $pm', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['config']['kernel'], d.get('parity_spot_check'))"
  done
done
