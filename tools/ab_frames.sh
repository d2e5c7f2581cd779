#!/bin/bash
# A/B of the batched frame call across library builds on one GPU box: tools/ab_frames.sh lib1.so lib2.so ...
for lib in "$@"; do
  VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python3 bench.py --legs frame --steps 5 --warmup 1 --strong-pairs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['batched_frames']
print('$lib', 'batch_ms %.3f' % d['ms_per_batch'], 'matcher_ms %.3f' % d['matcher_ms_per_batch'], 'fps %.0f' % d['frames_per_sec'])"
done
