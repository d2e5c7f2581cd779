#!/bin/bash
# The randomised campaigns with their logs kept: tools/run_fuzz_campaigns.sh <tag> [seconds each=240] [seed base=31] ["campaigns"=all]
# -> gpurun_out/fuzz_<tag>_{operators,chains,fast_solver,search,ragged,upkeep,shared}.log   (copy to profiles/<tag>_fuzz_*.log)
set -e
TAG=${1:-r05}
SEC=${2:-240}
SEED=${3:-31}
R=$PWD
mkdir -p $R/gpurun_out
i=0
ALL="operators chains fast_solver search ragged upkeep shared"
for t in $ALL; do
  i=$((i+1))
  case " ${4:-$ALL} " in *" $t "*) ;; *) continue;; esac
  echo "== fuzz_$t seed $((SEED+i)) for $SEC s"; date
  ( echo "# tools/fuzz_$t.py $((SEED+i)) $SEC   ($(date -u +%FT%TZ), $(python3 -c 'import sys; sys.path.insert(0, "."); import __graft_entry__ as g; v = g.load_package(); print(v.Context(0).device_info()[0])' 2>/dev/null))";
    python3 tools/fuzz_$t.py $((SEED+i)) $SEC ) > $R/gpurun_out/fuzz_${TAG}_$t.log 2>&1
  tail -1 $R/gpurun_out/fuzz_${TAG}_$t.log
done
