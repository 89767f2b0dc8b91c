#!/bin/bash
# one GPU pass of a round: the whole -m gpu suite, the default bench line, the headline profile (scripts/profile.sh) and the
# counter passes of the named non-headline cases
# usage: bash scripts/gpu_round.sh <tag> [profile-case ...]
tag=${1:-r04}; shift
out=gpurun_out/$tag; mkdir -p $out
if [ -z "$PAA_SKIP_TESTS" ]; then (timeout 900 python -m pytest tests -m gpu -q --no-header --durations=6 2>&1 | tail -16) > $out/tests.log; fi
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 900 bash scripts/profile.sh $tag > $out/profile.log 2>&1
python scripts/summarize_prof.py gpurun_out/prof_$tag gpurun_out/${tag}_fast800_w8_summary.json > $out/summarize.log 2>&1
rm -rf gpurun_out/prof_$tag/trace gpurun_out/prof_$tag/pmc1 gpurun_out/prof_$tag/pmc2 gpurun_out/prof_$tag/pmc3 gpurun_out/prof_$tag/pmc4      # (gpurun brings back at most 64 MiB)
for c in "$@"; do timeout 600 bash scripts/profile_kernel.sh $tag $c > $out/prof_$c.log 2>&1; done
# the device code this pass validated (copy to profiles/<round>_device_code.json when the suite was green)
python scripts/device_code_hash.py --write $out/device_code.json --note "pytest -m gpu + bench + profiles of gpurun_out/$tag" > /dev/null 2>&1
tail -12 $out/tests.log
