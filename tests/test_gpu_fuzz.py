"""Seeded slices (<= 20 s each) of the randomised campaigns (tools/fuzz_*.py): every operator and the reference-order
solver against the oracle bit for bit on random sizes / drop rates / motions / radii / thresholds; device-resident chains and
batches; the fast solver within its tolerances; the kd-tree's approximate modes and the exact radius search; round 5's map upkeep, open oneRound chains, device-side initialisation, the batched solver's helper waves.  The long
campaigns keep their logs under profiles/ (rNN_fuzz_*.log); these slices keep theirs under gpurun_out/."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed,seconds", [("fuzz_operators", 101, 20), ("fuzz_chains", 102, 20),
                                                ("fuzz_fast_solver", 103, 20), ("fuzz_search", 104, 15), ("fuzz_ragged", 105, 15),
                                                ("fuzz_upkeep", 106, 20), ("fuzz_shared", 107, 20)])
def test_fuzz_slice(tool, seed, seconds):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool + ".py"), str(seed), str(seconds)],
                       capture_output=True, text=True, timeout=600)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"fuzz_slice_{tool}_{seed}.log"), "w") as fh:
        fh.write(r.stdout + r.stderr)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "FAIL" not in r.stdout, r.stdout[-3000:]
    m = re.search(r"failures\s+(\d+)", r.stdout)
    assert m and int(m.group(1)) == 0, r.stdout[-2000:]
    n = [int(x) for x in re.findall(r"(?:iterations|sequences|batches|single problems|cases|matcher calls|frame calls|maps|chains|inits) (\d+)", r.stdout)]
    assert n and max(n) >= 5, r.stdout[-500:]                      # the slice really ran cases
