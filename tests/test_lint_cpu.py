"""CPU (compile-only): the ISA lint for the unconfirmed lane-mask hazard family (profiles/experiments/r02_lane_mask.md,
r02_square_branch.md).  No shipped kernel (render, backward, planes, dense) may contain a select on a lane mask that was
combined on the scalar unit from VALU compares (shape S1 of tools/lint_lane_masks.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_scalar_combined_lane_mask_selects_in_shipped_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_lane_masks.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    heads = [l for l in r.stdout.splitlines() if l.startswith("nfe_")]
    assert len(heads) == 4 and all("S1" in l for l in heads), r.stdout[:500]
    for l in heads:
        assert "S1 (select on scalar-combined VALU mask) = 0," in l, l
