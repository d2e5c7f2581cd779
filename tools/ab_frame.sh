#!/bin/bash
# A/B of the single-frame legs across library builds on one GPU box: tools/ab_frame.sh lib1.so lib2.so ...
for lib in "$@"; do
  VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python3 bench.py --legs frame --strong-pairs 0 --open-shares "" --frame-steps 200 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])['frame']
print('$lib', 'frames/s %.0f' % d['frames_per_sec'], 'match_ms %.4f' % d['match_ms'])"
done
