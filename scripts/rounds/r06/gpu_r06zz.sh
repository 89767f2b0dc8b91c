#!/bin/bash
# round 6, GPU pass zz (final build: real-input split of the 44 100- / 22 050- / 48 000- / 32 000- / 24 000-sample windows, kernels_wgs.hpp): the round's
# evidence pass (scripts/rounds/r06/gpu_round_r06.sh: whole -m gpu suite, bench line, headline profile, wave trace, device code), then kernel trace +
# counter passes of every non-headline case and the small kernels' trace
export TMPDIR=/tmp
tag=${1:-r06zz}
bash scripts/rounds/r06/gpu_round_r06.sh $tag
out=gpurun_out/$tag
for c in big_16000 big_16000_1h big_8000_batch big_44100 big_44100_20min big_22050 big_48000 big_32000 ct_400 ct_640_chromagram ct_640_spectrogram ct_640 ct_800_f64 ct_800_stereo fast_s800 mid_stats mix_256 mix_4800 \
         reg_chromagram_stereo reg_features_stereo reg_spectrogram_stereo w1024_spectrogram w1024 w1764 w1920 w2048 w2205 w2400 w512 w551_11k \
         blu_1103 blu_661 blu_736 blu_3002 blu_2203 blu_4001 blu_202 blu_1103_spectrogram; do
  timeout 300 bash scripts/profile_kernel.sh r06 $c 20 > $out/prof_$c.log 2>&1
  python -c "
import json
d=json.load(open('gpurun_out/r06_${c}_summary.json')); print('%-28s %-30s %8.1f us  traffic %s  conflicts %s  issue %s' % ('$c', d['run_under_trace']['kernel'], d.get('kernel_avg_us') or 0, d.get('traffic',{}).get('traffic_over_algorithmic'), d.get('lds_bank_conflict_ratio'), d.get('valu_issue_fraction')))"
done 2>&1 | tee $out/summary.txt
aux=$GRAFT_REPO_ROOT/gpurun_out/prof_r06_aux_kernels; mkdir -p $aux
rocprofv3 --kernel-trace --stats -d $aux/trace -o trace -- python scripts/aux_kernels_loop.py > $aux/run.log 2>&1
python scripts/summarize_aux_prof.py $aux gpurun_out/r06_aux_kernels_summary.json > $out/aux.log 2>&1
rm -rf $aux/trace
timeout 300 bash scripts/profile_similarity.sh r06 > $out/sim.log 2>&1
rm -rf gpurun_out/prof_r06_*/trace gpurun_out/prof_r06_*/pmc1 gpurun_out/prof_r06_*/pmc2 gpurun_out/prof_r06_*/pmc3 gpurun_out/prof_r06_*/pmc4
tail -3 $out/aux.log | cut -c1-200
