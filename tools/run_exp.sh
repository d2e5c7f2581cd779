for v in "$@"; do
  echo "== EXP $v"
  cd /tmp && export TMPDIR=/tmp
  VO_HIP_LIB=/root/repo/visual-odometry_amd/libvo_hip_exp$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_exp$v -- python3 /root/repo/tools/batch_frames.py 200 > /root/repo/gpurun_out/prof_exp$v.log 2>&1
  cd /root/repo
  python3 - <<PY
import csv, glob
f = glob.glob("/root/repo/gpurun_out/prof_exp$v/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f))):
    if "${KPAT:-cell_}" in r["Name"]: print("%-52s avg %9.1f us" % (r["Name"][:52], float(r["AverageNs"])/1e3))
PY
done
