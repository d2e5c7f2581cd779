#!/bin/bash
# What is kept under profiles/ for a round, in one call on the GPU box (a gpurun call is limited to 20 minutes; the randomised
# campaigns are their own call: tools/run_fuzz_campaigns.sh -- ONE run on the final kernels).   usage: tools/evidence.sh <tag>
#   bench line; kernel stats + FETCH_SIZE / WRITE_SIZE / VALU counter passes of bench.py (tools/collect_profiles.sh);
#   headline-only trace with its untraced twin (tools/prof_headline.sh); in-kernel phase stamps of the headline round
#   (tools/stamps_headline.sh); SQ counters of one 200-frame call (tools/pmc_frames.sh); DRAM-destined read requests of the
#   batched solver at 200 / 256 / 512 / 4096 problems (tools/pmc_dram.sh); counters of the matcher stage (tools/pmc_match.sh).
# Everything is summarised ON THE BOX into gpurun_out/<tag>_* (the raw rocprofv3 directories are deleted: a call may bring
# back 64 MiB) and every file it leaves carries the stamp of the sources it ran on (tools/stamp.py: commit from
# gpurun_stamp.json -- run `python tools/stamp.py write` in the build container before the gpurun call --, hash of csrc/,
# sha256 of the library).  Then, in the build container: cp gpurun_out/<tag>_* profiles/.
set -e
TAG=${1:-r05}
G=gpurun_out
python3 tools/stamp.py json > $G/${TAG}_stamp.json; python3 tools/stamp.py line
# counter passes and the in-kernel stamps FIRST, summarised into profiles/ on the box, so that the bench line quotes THIS run's
# summaries (bench.py names the file it reads and says whether its csrc_sha is the running tree's)
tools/collect_profiles.sh $TAG > $G/${TAG}_collect.log 2>&1
python3 tools/summarize_pmc.py $G/prof_$TAG $G/$TAG > /dev/null
rm -rf $G/prof_$TAG
tools/stamps_headline.sh $TAG | tail -12
cp $G/stamps_$TAG/stamps.txt $G/${TAG}_headline_stamps.txt; cp $G/stamps_$TAG/stamps.json $G/${TAG}_headline_stamps.json; rm -rf $G/stamps_$TAG
python3 tools/stamp.py embed $G/${TAG}_pmc_fetch_write.json $G/${TAG}_pmc_valu.json $G/${TAG}_headline_stamps.json
cp $G/${TAG}_pmc_fetch_write.json $G/${TAG}_pmc_valu.json $G/${TAG}_headline_stamps.json profiles/
python3 bench.py > $G/${TAG}_bench_line.json 2> $G/${TAG}_bench_stderr.log || (tail -20 $G/${TAG}_bench_stderr.log; exit 1)
tools/prof_headline.sh ${TAG}h 200 | tail -4
cp $G/prof_${TAG}h/check.txt $G/${TAG}_headline_check.txt; cp $G/prof_${TAG}h/stats/*/*kernel_stats.csv $G/${TAG}_headline_kernel_stats.csv; rm -rf $G/prof_${TAG}h
tools/pmc_frames.sh $TAG 200 > $G/${TAG}_sq_frames.txt
head -8 $G/${TAG}_sq_frames.txt; rm -rf $G/pmc_$TAG
tools/pmc_dram.sh $TAG | tail -6
cp $G/pmc_dram_$TAG/summary.txt $G/${TAG}_pmc_dram.txt; rm -rf $G/pmc_dram_$TAG
tools/pmc_match.sh $TAG > $G/${TAG}_pmc_match.txt 2>&1; head -12 $G/${TAG}_pmc_match.txt; rm -rf $G/pmc_match_$TAG
python3 tools/stamp.py embed $G/${TAG}_bench_line.json \
   $G/${TAG}_headline_check.txt $G/${TAG}_headline_stamps.txt $G/${TAG}_sq_frames.txt $G/${TAG}_pmc_dram.txt $G/${TAG}_pmc_match.txt
du -sh $G
