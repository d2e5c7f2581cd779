#!/bin/bash
# A/B of the headline on ONE GPU box: tools/ab_headline.sh [lib.so ...]  (default: the reference build against
# visual-odometry_amd/libvo_hip_exp.so, make -C csrc exp); three interleaved repetitions
set -e
LIBS=${@:-libvo_hip.so libvo_hip_exp.so}
for i in 1 2 3; do
  for lib in $LIBS; do
    VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python bench.py --no-extras --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib'.ljust(20), '%.0f iter/s  %.3f us/round  pose_err %.1e' % (d['value'], d['roofline']['launch_us'], d['pose_err_vs_gt']))"
  done
done
