#!/bin/bash
# SQ counters of the batched solver kernel (one --pmc pass): tools/pmc_batched.sh <tag>
set -e
TAG=${1:-sq}
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -- python3 $R/bench.py --legs batched --steps 5 --warmup 1 --strong-pairs 0 > $OUT.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "picp_batch_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 200 * 768:
        agg[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
        agg["dur"]["v"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print("%-22s %.4g  (n=%d, max %.4g)" % (k, sum(v["v"]) / len(v["v"]), len(v["v"]), max(v["v"])))
PY
