"""Replace `s_and_b64 D, A, B` + `v_cndmask_b32 vX, 0, vY, D` (D consumed once) by two chained selects on A and B - the scalar unit no
longer touches the masks, every other instruction of the compiler's failing assembly stays where it is."""
import re
from common import run

SAND = re.compile(r"^\s*s_and_b64\s+(vcc|s\[\d+:\d+\]),\s*(vcc|s\[\d+:\d+\]),\s*(vcc|s\[\d+:\d+\])\s*$")
CND = re.compile(r"^\s*v_cndmask_b32_e(32|64)\s+(v\d+),\s*0,\s*(v\d+),\s*(vcc|s\[\d+:\d+\])\s*$")
WRITES = re.compile(r"^\s*(s_\w+|v_cmp\w*|v_\w+)\s+(vcc|s\[\d+:\d+\])[, ]")
pending = {}      # line index of the select -> (A, B)
skip = set()


def plan(lines, start, end):
    for i in range(start, end):
        m = SAND.match(lines[i].split(";")[0].rstrip())
        if not m or "exec" in lines[i]:
            continue
        D, A, B = m.groups()
        for j in range(i + 1, min(i + 10, end)):
            lj = lines[j].split(";")[0].rstrip()
            c = CND.match(lj)
            if c and c.group(4) == D:
                pending[j] = (A, B)
                skip.add(i)
                break
            w = WRITES.match(lj)
            if w and w.group(2) in (D, A, B):      # a mask is redefined before the select: leave this site alone
                break
            if lj.strip().startswith(("s_cbranch", "s_branch")) or re.match(r"^[.\w$]+:", lj):
                break


planned = [False]


def edit(l, lines, i):
    if not planned[0]:
        from common import KERNEL
        start = next(k for k, x in enumerate(lines) if x.startswith(KERNEL))
        end = next(k for k in range(start, len(lines)) if "s_endpgm" in lines[k])
        plan(lines, start, end)
        planned[0] = True
    if i in skip:
        return ["\t; (s_and removed) " + l.strip()], 1
    if i in pending:
        A, B = pending[i]
        c = CND.match(l.split(";")[0].rstrip())
        dst, src = c.group(2), c.group(3)
        first = f"\tv_cndmask_b32_e32 {dst}, 0, {src}, vcc" if A == "vcc" else f"\tv_cndmask_b32_e64 {dst}, 0, {src}, {A}"
        second = f"\tv_cndmask_b32_e32 {dst}, 0, {dst}, vcc" if B == "vcc" else f"\tv_cndmask_b32_e64 {dst}, 0, {dst}, {B}"
        return [first, second], 0
    return [l], 0


run(edit)
