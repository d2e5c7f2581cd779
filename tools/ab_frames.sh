#!/bin/bash
# A/B of the whole batched frame call (tools/batch_frames.py, 200 x 50k) across library builds / environment settings on one
# GPU box: tools/ab_frames.sh "lib.so [ENV=val ...]" ...   e.g.  tools/ab_frames.sh "libvo_hip.so" "libvo_hip.so VO_JOIN_GATHER=0"
for spec in "$@"; do
  set -- $spec
  lib=$1; shift
  echo "== $lib $*"
  env "$@" VO_HIP_LIB=$PWD/visual-odometry_amd/$lib python3 tools/batch_frames.py ${F:-200} 2>/dev/null | grep -v amdgpu.ids
done
