#!/bin/bash
# kernel trace of the frame legs only (matcher / batched-frame tuning).  usage (GPU box): tools/prof_frame.sh <tag>
set -e
TAG=${1:-frame}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 5 --legs frame --gen-workers 1 --strong-pairs 0 > $OUT/stats.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-48s calls %5s total %8.3f ms avg %9.1f us max %9.1f" % (r["Name"].replace("void ","").replace("vo::","")[:48], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MaxNs"])/1e3))
PY
