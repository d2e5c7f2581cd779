#!/bin/bash
# kernel trace of tools/batch_frames.py (vo_frames_batch_dev at F frames).  usage (GPU box): tools/prof_batch.sh <tag> [F]
set -e
TAG=${1:-batch}
F=${2:-200}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 $R/tools/batch_frames.py $F > $OUT/plain.log 2>&1 || true
cat $OUT/plain.log | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/batch_frames.py $F > $OUT/stats.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows[:24]:
    per_batch = float(r["TotalDurationNs"]) / 11 / 1e6      # 11 runs of the batch (1 warm-up + 10 timed)
    tot += per_batch
    print("%-52s calls %5s avg %9.1f us  per batch %7.3f ms" % (r["Name"].replace("void ","").replace("vo::","")[:52], r["Calls"], float(r["AverageNs"])/1e3, per_batch))
print("sum per batch %.3f ms" % tot)
PY
