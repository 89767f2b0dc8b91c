#!/bin/bash
# round 5, GPU pass ap: the N > 1 path of the bench on the one-GPU box -- a bare `python bench.py --gpus 2 --no-gather` launches its
# own two ranks (both on device 0: no RCCL); and the bare `--gpus 2` (with the gather) must refuse loudly there
out=gpurun_out/r05ap; mkdir -p $out
timeout 600 python bench.py --gpus 2 --no-gather --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_n2_nogather.json 2> $out/n2.err
tail -1 $out/bench_n2_nogather.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'value', d['value'], 'ms', d['ms_per_step'], 'launched_by', d['config'].get('launched_by'), 'devices', d['config'].get('devices'), 'rccl_ranks', d['config'].get('rccl_ranks'), 'workload', d['config']['workload'][:60])"
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $out/bench_n2_gather.out 2> $out/n2_gather.err; echo "rc of the bare --gpus 2: $?"; tail -2 $out/n2_gather.err | cut -c1-300
