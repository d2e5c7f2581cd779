#!/bin/bash
# SQ counters of every kernel of one vo_frames_batch_dev call (tools/batch_frames.py): tools/pmc_frames.sh <tag> [F]
# two --pmc passes (the counters do not fit one); lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU * 4)
set -e
TAG=${1:-sqf}
F=${2:-200}
R=$PWD
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/a -- python3 $R/tools/batch_frames.py $F > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/b -- python3 $R/tools/batch_frames.py $F > $OUT/b.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").replace("vo::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("SQ_WAVES",):
                agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_ns", [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m.get("dur_ns", 0) < 20000: continue
    lanes = m.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * 4 * m.get("SQ_ACTIVE_INST_VALU", 0))
    print("%-40s %8.1f us  waves %8d  valu/wave %7.0f  lds/wave %6.0f  salu/wave %6.0f  vmem_rd/wave %5.0f  branch/wave %5.0f  lane util %.2f  valu busy %.2f  wait_inst %.2f  lds busy %.2f  bank conflict %.2f"
          % (k[:40], m["dur_ns"] / 1e3, m.get("SQ_WAVES", 0), m.get("SQ_INSTS_VALU", 0) / max(1, m.get("SQ_WAVES", 1)),
             m.get("SQ_INSTS_LDS", 0) / max(1, m.get("SQ_WAVES", 1)), m.get("SQ_INSTS_SALU", 0) / max(1, m.get("SQ_WAVES", 1)),
             m.get("SQ_INSTS_VMEM_RD", 0) / max(1, m.get("SQ_WAVES", 1)), m.get("SQ_INSTS_BRANCH", 0) / max(1, m.get("SQ_WAVES", 1)), lanes,
             m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(1.0, m.get("SQ_BUSY_CYCLES", 1)) , m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)),
             m.get("SQ_ACTIVE_INST_LDS", 0) * 4 / max(1.0, m.get("SQ_BUSY_CYCLES", 1)), m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_ACTIVE_INST_LDS", 1) )))
PY
