#!/bin/bash
# round 4, first GPU pass: headline profile (tracked as profiles/r04_*), the N = 2 socket-control-plane bench on one device,
# the big-window path's rate
out=gpurun_out/r04a; mkdir -p $out
bash scripts/profile.sh r04 > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_r04 gpurun_out/r04_fast800_w8_summary.json > $out/summarize.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --no-gather --clips 4000 --steps 20 --warmup 5 > $out/bench_n2_nogather.json 2> $out/bench_n2.err
timeout 300 python scripts/kernel_loop.py --case big_16000 --launches 5 --warmup 1 > $out/big.json 2>&1
tail -3 $out/bench_n2_nogather.json | cut -c1-600; tail -2 $out/bench_n2.err; cat $out/big.json | tail -2
