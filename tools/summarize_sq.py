#!/usr/bin/env python3
"""Per-kernel table from the two SQ counter passes of tools/pmc_frames.sh: tools/summarize_sq.py <gpurun_out/pmc_TAG> [> profiles/rNN_sq_frames.txt]
lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)  (1.00 = every lane active in every VALU instruction);
waiting = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of a wave's life spent waiting for an instruction's operands / memory)."""
import collections, csv, glob, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob("%s/%s/*/*counter_collection.csv" % (out, sub)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").replace("vo::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVES":
                agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("%-34s %9s %8s %10s %9s %9s %9s %9s %10s %8s" % ("kernel (200 frames x 50k per launch)", "us", "waves", "VALU/wave", "LDS/wave", "SALU/wave", "VMEMrd/w", "branch/w", "lane util", "waiting"))
for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_ns", [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if m.get("dur_ns", 0) < 15000:
        continue
    w = max(1.0, m.get("SQ_WAVES", 1))
    print("%-34s %9.1f %8d %10.0f %9.0f %9.0f %9.0f %9.0f %10.2f %8.2f" % (
        k[:34], m["dur_ns"] / 1e3, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SALU", 0) / w,
        m.get("SQ_INSTS_VMEM_RD", 0) / w, m.get("SQ_INSTS_BRANCH", 0) / w,
        m.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64 * m.get("SQ_ACTIVE_INST_VALU", 0)),
        m.get("SQ_WAIT_INST_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1))))
