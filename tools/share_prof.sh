#!/bin/bash
# kernel times of the batched solver under the settings of tools/share_ab.py (rocprofv3 --kernel-trace --stats per setting)
#   usage (GPU box): tools/share_prof.sh P "setting setting ..."      setting: VAR=VALUE[,VAR=VALUE]  (environment of the run)
cd /tmp && export TMPDIR=/tmp
P=${1:-200}
for st in ${2:-VO_PICP_SHARE=0 VO_PICP_SHARE=1}; do
  rm -rf /tmp/share_prof
  for kv in ${st//,/ }; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/share_prof -- python3 $GRAFT_REPO_ROOT/tools/share_ab.py child $P ${RAGGED:-0} > /tmp/share_prof.log 2>&1 || { tail -5 /tmp/share_prof.log; exit 1; }
  for kv in ${st//,/ }; do unset "${kv%%=*}"; done
  f=$(find /tmp/share_prof -name '*kernel_stats.csv' | head -1)
  echo "== $st (P=$P): $(grep '"ms"' /tmp/share_prof.log)"
  [ -n "$f" ] || { tail -20 /tmp/share_prof.log; find /tmp/share_prof | head; exit 1; }
  python3 -c 'import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "picp_batch" in r["Name"] or "fillBuffer" in r["Name"]:
        print("   %-58.58s calls %4s  avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))' "$f"
done
