#!/bin/bash
# A/B of the batched solver kernel across library builds on one GPU box: tools/ab_batched.sh lib1.so lib2.so ...
for lib in "$@"; do
  VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python3 bench.py --legs batched --steps 5 --warmup 1 --strong-pairs 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])['batched']
print('$lib', 'kernel_ms %.3f' % d['kernel_ms'], 'GB/s %.0f' % d['roofline']['achieved'], 'chip_full', [(c['pairs'], round(c['kernel_ms'],3)) for c in d.get('chip_full',[])])"
done
