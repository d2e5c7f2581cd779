#!/usr/bin/env python3
"""Build container, after tools/evidence.sh / evidence_extra.sh / run_fuzz_campaigns.sh came back through gpurun_out/:
copies the round's files into profiles/ and writes profiles/<tag>_stamp.json (the stamp of the run + the list of files it
covers: the .csv files cannot carry a stamp of their own).   usage: python tools/collect_into_profiles.py [tag=r05]"""
import glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
keep = ["bench_line.json", "bench_kernel_stats.csv", "headline_check.txt", "headline_kernel_stats.csv", "headline_stamps.json",
        "headline_stamps.txt", "pmc_dram.txt", "pmc_fetch_write.json", "pmc_match.txt", "pmc_valu.json", "sq_frames.txt",
        "one_round_api.txt", "nocopy_matcher.txt", "one_pass_ab.txt", "headline_per_thread.txt", "share_ab.txt", "share_prof.txt"]
for k in keep:
    src = os.path.join(G, f"{tag}_{k}")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{k}"))
for src in glob.glob(os.path.join(G, f"fuzz_{tag}_*.log")):
    shutil.copy(src, os.path.join(P, f"{tag}_fuzz_" + os.path.basename(src)[len(f"fuzz_{tag}_"):]))
st = json.load(open(os.path.join(G, f"{tag}_stamp.json")))
files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(P, f"{tag}_*")) if not f.endswith(f"{tag}_stamp.json"))
json.dump({"stamp": st, "files": files,
           "note": "every file of this round's evidence and the sources it was taken on (tools/stamp.py); the kernel_stats.csv files "
                   "carry no stamp of their own -- this list is theirs"}, open(os.path.join(P, f"{tag}_stamp.json"), "w"), indent=1)
print(len(files), "files;", st)
