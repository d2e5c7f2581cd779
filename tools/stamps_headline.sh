#!/bin/bash
# Evidence for the headline kernel that fits inside the step: in-kernel phase stamps (diagnostic build) and the untraced
# launch_us of the shipped library, on one box.   usage (GPU box): tools/stamps_headline.sh <tag>
TAG=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/stamps_$TAG
mkdir -p $OUT
[ -f visual-odometry_amd/libvo_hip_stamps.so ] || make -s -C visual-odometry_amd/csrc stamps > $OUT/make.log 2>&1     # (built in the build container: the .so travels)
python3 bench.py --no-extras --steps 200 > $OUT/line_untraced.json 2> $OUT/stderr.log
VO_HIP_LIB=$R/visual-odometry_amd/libvo_hip_stamps.so python3 tools/stamp_rounds.py 30 $OUT/stamps.json > $OUT/stamps.txt 2>> $OUT/stderr.log
python3 - <<PY >> $OUT/stamps.txt
import json
un = json.loads([l for l in open("$OUT/line_untraced.json") if l.startswith("{")][-1])
st = json.load(open("$OUT/stamps.json"))
print("shipped library, same box, untraced: ms_per_step %.1f us, roofline.launch_us %.3f us (value %.0f iter/s)" % (un["ms_per_step"] * 1e3, un["roofline"]["launch_us"], un["value"]))
print("stamps build round-to-round %.3f us = %.3f x the shipped library's launch_us" % (st["round_to_round_us_mean"], st["round_to_round_us_mean"] / un["roofline"]["launch_us"]))
PY
cat $OUT/stamps.txt
