#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
(timeout 600 python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -6) > $out/tests.log
PAA_HIP_LIBRARY=pyaudioanalysis_amd/libpaa_hip_timing.so timeout 300 python scripts/phase_timing_tri.py w2400 w2205 reg_features_stereo > $out/phases.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 50 > $out/bench.json 2> $out/bench.err
head -3 $out/tests.log; cat $out/phases.log; cut -c1-400 $out/bench.json
