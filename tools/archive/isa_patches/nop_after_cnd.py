"""WAR VALU read -> SALU write: pad behind every select that reads a lane mask from vcc / an SGPR pair."""
from common import PAD, reads_sgpr_mask, run
run(lambda l, L, i: ([l, PAD.rstrip("\n")], 1) if reads_sgpr_mask(l) else ([l], 0))
