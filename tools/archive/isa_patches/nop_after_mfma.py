"""MFMA result / operand hazards: pad behind every v_mfma."""
from common import PAD, run
run(lambda l, L, i: ([l, PAD.rstrip("\n")], 1) if l.strip().startswith("v_mfma") else ([l], 0))
