#!/bin/bash
# A/B of the matcher stage across library builds on one GPU box: tools/ab_match.sh lib1.so lib2.so ...  (QUICK=1: copies only)
for lib in "$@"; do
  echo "== $lib"
  VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python3 tools/match_modes.py 2>/dev/null | grep -v amdgpu.ids
done
