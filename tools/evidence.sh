#!/bin/bash
# What is kept under profiles/ for a round, in one call on the GPU box (a gpurun call is limited to 20 minutes; the randomised
# campaigns are their own call: tools/run_fuzz_campaigns.sh -- ONE run on the final kernels).   usage: tools/evidence.sh <tag>
#   bench line; kernel stats + FETCH_SIZE / WRITE_SIZE / VALU counter passes of bench.py (tools/collect_profiles.sh);
#   headline-only trace with its untraced twin (tools/prof_headline.sh); in-kernel phase stamps of the headline round
#   (tools/stamps_headline.sh); SQ counters of one 200-frame call (tools/pmc_frames.sh); DRAM-destined read requests of the
#   batched solver at 200 / 256 / 512 problems (tools/pmc_dram.sh); counters of the matcher stage (tools/pmc_match.sh).
# Then, in the build container: tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<tag>, copy the rest into profiles/.
# Every file it leaves carries the stamp of the sources it ran on (tools/stamp.py: commit from gpurun_stamp.json -- run
# `python tools/stamp.py write` in the build container before the gpurun call --, hash of csrc/, sha256 of the library);
# ${TAG}_stamp.json keeps it for the summaries made afterwards (tools/stamp.py embed --from gpurun_out/${TAG}_stamp.json profiles/...).
set -e
TAG=${1:-r05}
python3 tools/stamp.py json > gpurun_out/${TAG}_stamp.json; python3 tools/stamp.py line
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log || (tail -20 gpurun_out/${TAG}_bench_stderr.log; exit 1)
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
tools/prof_headline.sh ${TAG}h 200 | tail -4
tools/stamps_headline.sh $TAG | tail -12
tools/pmc_frames.sh $TAG 200 > gpurun_out/${TAG}_sq_frames.txt
head -8 gpurun_out/${TAG}_sq_frames.txt
tools/pmc_dram.sh $TAG | tail -5
tools/pmc_match.sh $TAG > gpurun_out/${TAG}_pmc_match.txt 2>&1; head -12 gpurun_out/${TAG}_pmc_match.txt
python3 tools/stamp.py embed gpurun_out/${TAG}_bench_line.json gpurun_out/${TAG}_sq_frames.txt gpurun_out/${TAG}_pmc_match.txt
