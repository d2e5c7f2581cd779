#!/bin/bash
# How much of the batched solver's read traffic goes to DRAM and how much is answered by the Infinity Cache: one --pmc
# pass with the L2's memory-side read requests (all / DRAM-destined), one with FETCH_SIZE, over bench.py --legs batched
# (200, 256 and 512 problems).   usage (GPU box): tools/pmc_dram.sh <tag>
TAG=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/pmc_dram_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/a -- python3 $R/bench.py --legs batched --steps 5 --warmup 1 --strong-pairs 0 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -- python3 $R/bench.py --legs batched --steps 5 --warmup 1 --strong-pairs 0 > $OUT/b.log 2>&1
cd $R
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "picp_batch_kernel" not in r["Kernel_Name"]: continue
        g = int(r["Grid_Size"]) // 768
        agg[g][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[g]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("picp_batch_kernel, 50k correspondences x 50 rounds per problem (launches with all rounds only):")
for g in sorted(agg):
    d = agg[g]
    full = [i for i, t in enumerate(d["dur_us"]) if t > 0.5 * max(d["dur_us"])]
    def m(k):
        v = d[k]
        sel = [v[i] for i in range(len(v)) if len(v) != len(d["dur_us"]) or i in full]
        big = [x for x in v if x > 0.5 * max(v)] if v else []
        return sum(big) / len(big) if big else float("nan")
    rd, dram, fetch = m("TCC_EA0_RDREQ_sum"), m("TCC_EA0_RDREQ_DRAM_sum"), m("FETCH_SIZE")
    print("  %4d problems: working set %6.1f MB  RDREQ %.4g  of them DRAM-destined %.4g (%.1f %%)  FETCH_SIZE %.4g KB  (x2 for 16-B/lane loads = %.3f GB)  dur %.0f us"
          % (g, g * 1.0, rd, dram, 100.0 * dram / rd if rd else float("nan"), fetch, 2 * fetch * 1024 / 1e9, max(d["dur_us"])))
PY
