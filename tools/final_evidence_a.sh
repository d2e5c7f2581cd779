#!/bin/bash
# First half of tools/final_evidence.sh (a gpurun call is limited to 20 minutes): bench line, kernel stats + counter passes,
# headline-only trace with its untraced twin, SQ counters of one 200-frame call.  usage: tools/final_evidence_a.sh <tag>
set -e
TAG=${1:-r03}
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log || (tail -20 gpurun_out/${TAG}_bench_stderr.log; exit 1)
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
tools/prof_headline.sh ${TAG}h 200 | tail -4
tools/pmc_frames.sh $TAG 200 > gpurun_out/${TAG}_sq_frames.txt
head -8 gpurun_out/${TAG}_sq_frames.txt
