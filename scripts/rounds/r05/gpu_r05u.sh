#!/bin/bash
# round 5, GPU pass u: launch geometry of the hot kernel (scripts/experiments/run_geometry.py): is filling the last six CUs worth 2 %?
mkdir -p gpurun_out/r05u
timeout 600 python scripts/experiments/run_geometry.py > gpurun_out/r05u/run_geometry.json 2> gpurun_out/r05u/err.log
cat gpurun_out/r05u/run_geometry.json; tail -3 gpurun_out/r05u/err.log
