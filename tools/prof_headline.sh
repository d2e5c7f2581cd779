#!/bin/bash
# Kernel trace of the HEADLINE measurement alone (bench.py --no-extras): every launch in the trace belongs to the timed
# geometry (one 50k pair, 50 rounds per step), so 50 x the round kernel's average must fit the same run's ms_per_step.
# usage (GPU box): tools/prof_headline.sh <tag> [steps]     ->  gpurun_out/prof_<tag>/{stats/,line.json,check.txt}
set -e
TAG=${1:-headline}
STEPS=${2:-200}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --no-extras --steps $STEPS > $OUT/line.json 2> $OUT/stderr.log
python3 $R/bench.py --no-extras --steps $STEPS > $OUT/line_untraced.json 2>> $OUT/stderr.log     # the same command without the tracer
cd $R
python3 - <<PY | tee $OUT/check.txt
import csv, glob, json
line = json.loads([l for l in open("$OUT/line.json") if l.startswith("{")][-1])
f = glob.glob("$OUT/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:8]:
    print("%-60s calls %6s avg %9.1f ns" % (r["Name"].replace("void ", "")[:60], r["Calls"], float(r["AverageNs"])))
k = [r for r in rows if "picp_round_kernel<true, false, true, false>" in r["Name"]][0]
avg_us = float(k["AverageNs"]) / 1e3
it = line["config"]["iters_per_step"]
print("round kernel average %.3f us x %d = %.1f us ; ms_per_step (same run) %.1f us ; roofline.launch_us %.3f (%.1f %% off the trace average)"
      % (avg_us, it, avg_us * it, line["ms_per_step"] * 1e3, line["roofline"]["launch_us"], 100 * (line["roofline"]["launch_us"] / avg_us - 1)))
print("value under the tracer %.0f iter/s" % line["value"])
un = json.loads([l for l in open("$OUT/line_untraced.json") if l.startswith("{")][-1])
print("untraced run of the same command: value %.0f iter/s, ms_per_step %.1f us, roofline.launch_us %.3f (%.1f %% off the trace average of the kernel)"
      % (un["value"], un["ms_per_step"] * 1e3, un["roofline"]["launch_us"], 100 * (un["roofline"]["launch_us"] / avg_us - 1)))
print("(the tracer adds ~2.5 us of host/dispatch overhead per launch to the step: event times of a traced run are not kernel times)")
PY
