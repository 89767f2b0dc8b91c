#!/bin/bash
# round 5, GPU pass z: wave trace of the hot kernel (-DPAA_F800_TRACE build) after 3 s of untimed launches -- pass y's trace ran
# at 2.08 GHz after 0.5 s on its box while the bench of the same pass implies the 2.2 GHz of round 4's trace
out=gpurun_out/r05z; mkdir -p $out
for pre in 3 6; do
(echo "# scripts/phase_timing.py with the -DPAA_F800_TRACE build (per-wave life times only), headline plan (1-hour clip, 800/400), after $pre s of untimed launches"
 PAA_PHASE_PREWARM=$pre PAA_HIP_LIBRARY=$PWD/pyaudioanalysis_amd/libpaa_hip_trace.so timeout 300 python scripts/phase_timing.py | grep -v "0.00 %") > $out/wave_trace_$pre.txt 2> $out/wave_trace_$pre.err
sed -n 1,8p $out/wave_trace_$pre.txt
done
