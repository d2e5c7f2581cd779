#!/bin/bash
# kernel trace of the reference-order solver (tools/exact_rate.py, 50k correspondences) and of the small-problem form (tools/small_rate.py, 127)
set -e
R=$PWD
OUT=$R/gpurun_out/prof_exact
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/exact -- python3 $R/tools/exact_rate.py > $OUT/exact.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/small -- python3 $R/tools/small_rate.py 127 1000 > $OUT/small.log 2>&1
cd $R
tail -2 $OUT/exact.log; tail -1 $OUT/small.log
head -4 $OUT/exact/*/*kernel_stats.csv | cut -c1-200
head -4 $OUT/small/*/*kernel_stats.csv | cut -c1-200
