#!/bin/bash
# kernel trace of the matcher stage with a share of open queries in every frame: tools/prof_open.sh <share> [tag]
set -e
SHARE=${1:-0.01}
TAG=${2:-open}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
OPENS=$SHARE rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/match_modes.py > $OUT/stats.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:26]:
    print("%-44s calls %5s total %8.3f ms avg %9.1f us min %8.1f max %9.1f" % (r["Name"].replace("void ","").replace("vo::","")[:44], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
