#!/bin/bash
# HBM-side counters of the matcher stage's kernels (tools/match_modes.py, QUICK=1): separate --pmc passes for FETCH_SIZE,
# WRITE_SIZE and the L2 hit / miss counts.  usage (GPU box): tools/pmc_match.sh <tag>
TAG=${1:-m}
R=$PWD
OUT=$R/gpurun_out/pmc_match_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$OUT/$(echo $C | tr ' ' '+')
  QUICK=1 F=${F:-200} timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 $R/tools/match_modes.py > $D.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vo::", "")
        agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[(name, int(r["Grid_Size"]))]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), d in sorted(agg.items(), key=lambda kv: -max(kv[1]["dur_us"])):
    if max(d["dur_us"]) < 20: continue
    print("%-40s grid %9d  " % (name[:40], grid) + "  ".join("%s %.4g" % (k, sum(v) / len(v)) for k, v in sorted(d.items())))
PY
