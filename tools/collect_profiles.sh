#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + separate PMC passes of bench.py.
# Output: gpurun_out/prof_<tag>/{stats,FETCH_SIZE,WRITE_SIZE}/...   then: python tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<tag>
# The config-4 pairs are generated in-process (--gen-workers 1: no forked helpers under the profiler).
set -e
TAG=${1:-r05}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "stats pass"; date
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 50 --warmup 5 --legs frame,batched,sequence --strong-pairs 0 --open-shares "" --gen-workers 1 > $OUT/stats.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  echo "$C pass"; date
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$C -- python3 $R/bench.py --steps 5 --warmup 1 --legs frame,batched --frame-steps 3 --strong-pairs 0 --open-shares "" --gen-workers 1 > $OUT/$C.log 2>&1
done
# third counter pass: executed VALU instructions (the matcher's roofline is the FP32 vector peak)
echo "VALU pass"; date
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/VALU -- python3 $R/bench.py --steps 5 --warmup 1 --legs frame,batched --frame-steps 3 --strong-pairs 0 --open-shares "" --gen-workers 1 > $OUT/VALU.log 2>&1
cd $R
echo done; date; ls $OUT
