"""VALU write -> MFMA read hazards: pad in front of every v_mfma."""
from common import PAD, run
run(lambda l, L, i: ([PAD.rstrip("\n"), l], 1) if l.strip().startswith("v_mfma") else ([l], 0))
