#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun); summaries land in gpurun_out/prof_<tag>/
tag=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-extras > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out/pmc1 -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $out/pmc2 -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o pmc -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_pmc4.log 2>&1
find $out -name "*.csv" | head -30
ls -la $out/*
