#!/bin/bash
# rocprofv3 passes of ONE non-headline kernel (scripts/kernel_loop.py --case <case>): kernel trace + statistics, then
# separate counter passes (issue counters x2, FETCH_SIZE, WRITE_SIZE -- never combined with tracing).
# usage (on the GPU box through gpurun): bash scripts/profile_kernel.sh <tag> <case> [launches]
tag=$1; case=$2; n=${3:-100}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_${case}
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python scripts/kernel_loop.py --case $case --launches $n > $out/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $out/pmc1 -o pmc -- python scripts/kernel_loop.py --case $case --launches 4 --warmup 1 > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $out/pmc2 -o pmc -- python scripts/kernel_loop.py --case $case --launches 4 --warmup 1 > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o pmc -- python scripts/kernel_loop.py --case $case --launches 4 --warmup 1 > $out/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o pmc -- python scripts/kernel_loop.py --case $case --launches 4 --warmup 1 > $out/bench_pmc4.log 2>&1
python scripts/summarize_kernel_prof.py $out $GRAFT_REPO_ROOT/gpurun_out/${tag}_${case}_summary.json
# (the raw rocpd databases are tens of MB per case and gpurun brings back at most 64 MiB: keep the summary and the logs)
rm -rf $out/trace $out/pmc1 $out/pmc2 $out/pmc3 $out/pmc4
