"""SALU -> VALU: pad in front of every select that reads a lane mask from vcc / an SGPR pair."""
from common import PAD, reads_sgpr_mask, run
run(lambda l, L, i: ([PAD.rstrip("\n"), l], 1) if reads_sgpr_mask(l) else ([l], 0))
