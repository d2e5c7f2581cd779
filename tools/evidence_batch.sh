#!/bin/bash
# The batched-path half of tools/evidence.sh in an order that needs one call: counter passes first, their summaries written
# into profiles/ ON THE BOX (bench.py quotes them), then the bench line, then the SQ table and the matcher's counters;
# the summaries travel back through gpurun_out/.   usage (GPU box): tools/evidence_batch.sh [tag=r04]
set -e
TAG=${1:-r04}
tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
python3 tools/summarize_pmc.py gpurun_out/prof_$TAG profiles/$TAG > /dev/null
cp profiles/${TAG}_pmc_fetch_write.json profiles/${TAG}_pmc_valu.json profiles/${TAG}_bench_kernel_stats.csv gpurun_out/ 2>/dev/null || true
echo collect done
python3 bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log || (tail -20 gpurun_out/${TAG}_bench_stderr.log; exit 1)
echo bench done
tools/pmc_frames.sh $TAG 200 > gpurun_out/${TAG}_sq_frames.txt
head -12 gpurun_out/${TAG}_sq_frames.txt
tools/pmc_match.sh $TAG > gpurun_out/${TAG}_pmc_match.txt 2>&1; head -6 gpurun_out/${TAG}_pmc_match.txt
