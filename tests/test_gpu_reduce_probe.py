"""GPU test (-m gpu): the team additions of the bucket reduction (csrc/curve.h team2_add / team4_add with the DPP exchange the
library uses) against the one-lane addition ON THE DEVICE, operand pair by operand pair -- general pairs, equal and opposite points,
infinity on either side, G1 and G2 -- through tools/build/reduce_probe (built by __graft_entry__.build()).  The MSM parity tests
cover the same code end to end; this one fails on the addition itself, which is where a wrong wait state in the cross-lane moves
showed first (profiles/r05_experiments.md section 1)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_team_additions_equal_the_one_lane_addition_on_the_device():
    exe = os.path.join(ROOT, "tools", "build", "reduce_probe")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tools"), "reduce_probe"])
    res = subprocess.run([exe, "check"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("the library's exchange")]
    assert len(lines) == 2, res.stdout                                   # G1 and G2
    for ln in lines:
        m = re.search(r"two lanes (\d+) wrong, four lanes (\d+) wrong \((\d+) sums are infinity\)", ln)
        assert m and m.group(1) == "0" and m.group(2) == "0" and int(m.group(3)) > 0, ln
