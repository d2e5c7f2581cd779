#!/bin/bash
# per-kernel averages of one batched frame call (tools/batch_frames.py 200) across library builds on one GPU box:
#   tools/ab_kernels.sh <kernel name pattern> lib1.so lib2.so ...
PAT=$1; shift
R=$PWD
for lib in "$@"; do
  OUT=$R/gpurun_out/abk_$lib
  rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && VO_HIP_LIB=$R/visual-odometry_amd/$lib REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/batch_frames.py 200 > $OUT.log 2>&1 )
  echo "== $lib  $(grep 'ms per batch' $OUT.log | sed 's/.*F=200: //;s/ =.*//')"
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in "$PAT".split("|")): print("   %-50s avg %8.1f us" % (r["Name"].replace("void ","").replace("vo::","")[:50], float(r["AverageNs"])/1e3))
PY
  rm -rf $OUT
done
