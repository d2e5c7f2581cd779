#!/bin/bash
# per-kernel VGPR / scratch / LDS / occupancy of the library's translation units (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../visual-odometry_amd/csrc"
for f in ${@:-match geom picp}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize \
      -mllvm -amdgpu-kernarg-preload-count=16 $EXTRA -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$f.o 2>&1 |
  python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-70s vgpr %4s sgpr %4s scratch %5s occ %2s lds %6s" % (cur["name"][:70], cur.get("VGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
'
done
