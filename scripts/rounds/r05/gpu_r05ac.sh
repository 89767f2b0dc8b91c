#!/bin/bash
# round 5, GPU pass ac: kernel trace of the small kernels no bench shape runs on its own (scripts/aux_kernels_loop.py)
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r05_aux_kernels; mkdir -p $out
timeout 600 python scripts/aux_kernels_loop.py > $out/plain.log 2>&1; tail -2 $out/plain.log | cut -c1-300
rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- python scripts/aux_kernels_loop.py > $out/run.log 2>&1
python scripts/summarize_aux_prof.py $out gpurun_out/r05_aux_kernels_summary.json | head -60
rm -rf $out/trace
