"""Insert / replace instructions at kernel-relative line numbers of bwd_scatter_sorted_kernel<true, true> (line 1 = its label).
    at_lines.py in.s out.s  <n>|<expect-substring>|<before,after,replace>|<text with ; for newlines>  ...
The expected substring guards against a listing that moved."""
import sys
from common import KERNEL

src, dst, specs = sys.argv[1], sys.argv[2], sys.argv[3:]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
edits = {}
if "vgpr256" in specs:          # the patch uses v255: the kernel descriptor must say so
    specs.remove("vgpr256")
    k = next(i for i in range(start, len(lines)) if ".amdhsa_next_free_vgpr 255" in lines[i])
    lines[k] = lines[k].replace("255", "256")
if "sgpr102" in specs:          # the patch uses s[100:101]
    specs.remove("sgpr102")
    k = next(i for i in range(start, len(lines)) if ".amdhsa_next_free_sgpr 100" in lines[i])
    lines[k] = lines[k].replace("100", "102")
for s in specs:
    n, expect, where, text = s.split("|", 3)
    i = start + int(n) - 1
    assert expect in lines[i], (n, expect, lines[i])
    edits[i] = (where, ["\t" + t.strip() for t in text.split(";") if t.strip()])
out = []
for i, l in enumerate(lines):
    if i in edits:
        where, new = edits[i]
        out += new + [l] if where == "before" else ([l] + new if where == "after" else new)
    else:
        out.append(l)
open(dst, "w").write("\n".join(out))
print(f"patched {len(edits)} lines")
