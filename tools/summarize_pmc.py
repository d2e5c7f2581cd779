#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh into the summaries kept under profiles/.
usage: python tools/summarize_pmc.py gpurun_out/prof_<tag> profiles/<prefix>"""
import collections
import os
import csv
import glob
import json
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]


def newest(pattern):
    """gpurun merges a call's files into what earlier calls left under gpurun_out/: take the latest run's file"""
    return max(glob.glob(pattern), key=os.path.getmtime)


WIDE = ("vo::picp_batch_kernel",)       # kernels whose streaming loads are 16 B per lane (dwordx4)
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(f"{src}/{C}/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[(name, int(r["Grid_Size"]))].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    for (k, grid), v in agg.items():
        if not k.startswith("vo::"):
            continue
        # one entry per (kernel, grid size in threads): the same kernel runs at several problem sizes in one bench
        d = out.setdefault(k, {"wide_16B_loads": k.startswith(WIDE), "by_grid": {}})["by_grid"].setdefault(str(grid), {})
        d[C + "_KB_mean"] = sum(x[0] for x in v) / len(v)
        d[C + "_KB_max"] = max(x[0] for x in v)     # bench.py also launches the batch kernel with 0 rounds (gather-pass timing)
        d["launches_" + C] = len(v)
        d["dur_us_under_pmc"] = sum(x[1] for x in v) / len(v) / 1e3
# one whole vo_frames_batch_dev call (config 4's per-GPU share): the kernels from cell_bounds_kernel over 200 frames to
# the tri_scatter_kernel that ends the call, summed (last complete call of the pass)
call = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(f"{src}/{C}/*/*counter_collection.csv")
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    cur, last = None, None
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        # a call over 200 x 50k frames starts with the matcher's first kernel: the row hashes of the exact-duplicate pass
        # (8 XCD classes x 25 frames x 196 workgroups of 256 threads), or -- VO_MATCH_HASH=0 -- the sampled bounds
        if (name == "vo::hash_rows_kernel" and int(r["Grid_Size"]) == 8 * 25 * 196 * 256) or \
           (name == "vo::cell_bounds_kernel" and int(r["Grid_Size"]) == 200 * 1024 and cur is None):
            cur = collections.OrderedDict()
        if cur is not None and name.startswith("vo::"):
            cur[name] = cur.get(name, 0.0) + float(r["Counter_Value"])
            if name == "vo::tri_scatter_kernel":
                # (the bench also runs the matcher chain of the batch alone: only a sequence that went through the solver is a call)
                if "vo::picp_batch_kernel<true, false>" in cur:
                    last = cur
                cur = None
    if last:
        call[C + "_KB_by_kernel"] = last
        call[C + "_KB"] = sum(last.values())
if call:
    wide_fetch = call.get("FETCH_SIZE_KB_by_kernel", {}).get("vo::picp_batch_kernel<true, false>", 0.0)
    call["bytes_corrected"] = (call.get("FETCH_SIZE_KB", 0.0) + wide_fetch + call.get("WRITE_SIZE_KB", 0.0)) * 1024.0
    call["note"] = ("sum over the kernels of one vo_frames_batch_dev call (200 frames x 50k); FETCH_SIZE doubled for the "
                    "batched solver (16-B/lane streaming loads), counted as reported for the rest (mixed widths: uncalibrated)")
json.dump({"batched_frames_call": call, "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 5 "
                      "--warmup 1 --legs frame,batched --frame-steps 3 --strong-pairs 0 --open-shares "" --gen-workers 1   (one pass per counter; tools/collect_profiles.sh)",
           "units": "FETCH_SIZE / WRITE_SIZE are KB.  gfx950: FETCH_SIZE counts exactly half the bytes of wide (16 B/lane) "
                    "coalesced streaming reads (MI355X_MICROARCH.md, HBM) -> doubled where wide_16B_loads is true; other "
                    "access widths are uncalibrated and reported as counted",
           "kernels": out}, open(dst + "_pmc_fetch_write.json", "w"), indent=1, sort_keys=True)
# executed VALU instructions per launch (third pass of tools/collect_profiles.sh): the matcher kernels are VALU-bound by design
vf = glob.glob(f"{src}/VALU/*/*counter_collection.csv")
if vf:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(vf, key=os.path.getmtime))):       # (the latest run's file, like newest())
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not name.startswith("vo::"):
            continue
        key = (name, int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVES":
            agg[key]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    valu = {}
    for (k, grid), c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        valu.setdefault(k, {"by_grid": {}})["by_grid"][str(grid)] = {
            "valu_insts_per_launch": m.get("SQ_INSTS_VALU", 0.0), "waves": m.get("SQ_WAVES", 0.0),
            "lane_utilisation": m.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0)),
            "dur_us_under_pmc": m.get("dur_ns", 0.0) / 1e3, "launches": len(c.get("SQ_WAVES", []))}
        d = valu[k]["by_grid"][str(grid)]
        # wave-instructions x 64 lanes x 2 flop (every instruction counted as an FMA) per second against 157.3 TFLOP/s
        d["valu_tflops_equiv_under_pmc"] = d["valu_insts_per_launch"] * 128.0 / max(d["dur_us_under_pmc"], 1e-9) / 1e6
        d["valu_frac_of_fp32_peak_under_pmc"] = d["valu_tflops_equiv_under_pmc"] / 157.3
    json.dump({"command": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES "
                          "--output-format csv -- python3 bench.py --steps 5 --warmup 1 --cpu-seconds 0 --frame-steps 3 --seq-frames 6 "
                          "--seq-points 5000 --strong-pairs 0 --gen-workers 1   (tools/collect_profiles.sh, third pass)",
               "units": "valu_insts_per_launch = SQ_INSTS_VALU (wave-instructions, summed over all waves of a launch); "
                        "tflops_equiv = insts x 64 lanes x 2 flop / kernel time: what the kernel would deliver if every VALU "
                        "instruction were an FMA with all lanes active -- the share of the FP32 vector peak (157.3 TFLOP/s, "
                        "MI355X_MICROARCH.md) its instruction stream occupies",
               "kernels": valu}, open(dst + "_pmc_valu.json", "w"), indent=1, sort_keys=True)
shutil.copy(newest(f"{src}/stats/*/*kernel_stats.csv"), dst + "_bench_kernel_stats.csv")
for k in ("vo::picp_batch_kernel<true, false>", "vo::picp_round_kernel<true, false, true, false>"):
    if k in out:
        print(k, json.dumps(out[k]))
