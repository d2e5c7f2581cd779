#!/bin/bash
# Everything kept under profiles/ for a round, in one call on the GPU box: tools/final_evidence.sh <tag> [fuzz seconds each]
#   bench line, kernel stats + three counter passes of bench.py, headline-only trace with its untraced twin, SQ counters of one
#   200-frame call, the four randomised campaigns.  Then (in the build container): tools/summarize_pmc.py, copy into profiles/.
set -e
TAG=${1:-r03}
FZ=${2:-150}
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log || (tail -20 gpurun_out/${TAG}_bench_stderr.log; exit 1)
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
tools/prof_headline.sh ${TAG}h 200 | tail -4
tools/pmc_frames.sh $TAG 200 > gpurun_out/${TAG}_sq_frames.txt
head -8 gpurun_out/${TAG}_sq_frames.txt
tools/run_fuzz_campaigns.sh ${TAG}c $FZ 91
