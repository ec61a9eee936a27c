"""RAW VALU -> SALU: pad in front of every scalar combination of lane masks."""
from common import PAD, is_mask_logic, run
run(lambda l, L, i: ([PAD.rstrip("\n"), l], 1) if is_mask_logic(l) else ([l], 0))
