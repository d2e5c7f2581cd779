#!/bin/bash
# Round 5's additions to tools/evidence.sh, one call on the GPU box:   tools/evidence_extra.sh [tag=r05]
#   the reference's own oneRound loop (apps/one_round_rate, tools/micro/host_call_cost);
#   kernel trace of the matcher stage when no query has a bitwise copy (tools/prof_open.sh 1.0);
#   single-pass against count/scan/scatter compaction on one box (VO_ONE_PASS, tools/ab_frames.sh + kernel trace);
#   the headline with 1..3 correspondences per thread (VO_PICP_PER_THREAD);
#   the batched solver with and without helper waves over the problem count (tools/share_ab.py), the hand-over's round trip
#   (tools/micro/hop_latency).
set -e
TAG=${1:-r05}
O=gpurun_out/${TAG}_one_round_api.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# tools/micro/bin/host_call_cost ; apps/bin/one_round_rate (x3)";
  tools/micro/bin/host_call_cost; for i in 1 2 3; do apps/bin/one_round_rate; done; } > $O 2>/dev/null
tail -4 $O
O=gpurun_out/${TAG}_nocopy_matcher.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# tools/prof_open.sh 1.0 : matcher stage of 200 x 50k frames, every current descriptor displaced (no bitwise copies), modes 3 and 5";
  tools/prof_open.sh 1.0 ${TAG}nocopy; grep batched gpurun_out/prof_${TAG}nocopy/stats.log; } > $O 2>&1
head -12 $O
O=gpurun_out/${TAG}_one_pass_ab.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# tools/ab_frames.sh: whole vo_frames_batch_dev call, 200 x 50k, count/scan/scatter (default) against the single-pass chained scan (VO_ONE_PASS=1)";
  tools/ab_frames.sh "libvo_hip.so" "libvo_hip.so VO_ONE_PASS=1" "libvo_hip.so" "libvo_hip.so VO_ONE_PASS=1";
  VO_ONE_PASS=1 tools/prof_batch.sh ${TAG}onepass 200 | grep -E "onepass|tri_|join_|sum per"; tools/prof_batch.sh ${TAG}twopass 200 | grep -E "tri_|join_|sum per"; } > $O 2>&1
cat $O | grep -v amdgpu.ids
O=gpurun_out/${TAG}_headline_per_thread.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# bench.py --no-extras --steps 400 with VO_PICP_PER_THREAD = 1, 2, 3 (correspondences per thread of the round kernel), twice";
  for i in 1 2; do for pt in 1 2 3; do echo -n "per_thread $pt: "; VO_PICP_PER_THREAD=$pt python3 bench.py --no-extras --steps 400 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f iter/s  %.3f us/round' % (d['value'], d['roofline']['launch_us']))"; done; done; } > $O
cat $O
O=gpurun_out/${TAG}_share_ab.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# tools/micro/bin/hop_latency: round trip of 32 tagged words between workgroup 0 and workgroup <key> of one launch (us)";
  tools/micro/bin/hop_latency 2000;
  echo "# tools/share_ab.py: vo_picp_solve_batch_dev, one workgroup per problem, 50k correspondences x 50 rounds: alone (share=0) / with helper waves (share=1)";
  RAGGED=0 python3 tools/share_ab.py 16,32,64,96,128,160,176,200 share=0 share=1; RAGGED=1 python3 tools/share_ab.py 64,128,200 share=0 share=1; } > $O 2>&1
cat $O
O=gpurun_out/${TAG}_share_prof.txt
{ echo "# $(python3 tools/stamp.py line)"; echo "# tools/share_prof.sh: rocprofv3 --kernel-trace --stats of tools/share_ab.py child (7 calls of 50 rounds, 50k correspondences per problem), alone / with helper waves";
  tools/share_prof.sh 64 "VO_PICP_SHARE=0 VO_PICP_SHARE=1"; tools/share_prof.sh 128 "VO_PICP_SHARE=0 VO_PICP_SHARE=1"; } > $O 2>&1
cat $O
rm -rf gpurun_out/prof_${TAG}nocopy gpurun_out/prof_${TAG}onepass gpurun_out/prof_${TAG}twopass gpurun_out/share_ab_T.npy
du -sh gpurun_out
