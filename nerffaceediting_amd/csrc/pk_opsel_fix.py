#!/usr/bin/env python3
"""Assembly pass of the build: removes an operand form of the packed-fp32 instructions that reads wrong data on MI355X.

Finding (round 4, profiles/experiments/r04_pk_opsel_hazard.md, tools/microbench/pk_opsel.hip): `v_pk_mul_f32 / v_pk_add_f32 /
v_pk_fma_f32` whose LOW result takes the LOW register of src0 and the HIGH register of a DIFFERENT src1 pair (`op_sel:[0,1]`,
`op_sel:[0,1,x]`) reads 0.0 for src1 in lanes 48..63 now and then (3e-5 per executed instruction in the microbenchmark) while
another wave of the SIMD has MFMAs in flight.  Silent, run dependent, no fault - the two-round-old "lanes 48-63" finding.  The
commuted encoding (`op_sel:[1,0]`: src0 high, src1 low), the same-pair form and every other op_sel combination tested never fail
(5e10 lane results each).  hipcc's SLP vectoriser emits the bad form for scalar code like `a0 * b1`, so it cannot be avoided
reliably at source level: this pass swaps src0 and src1 (multiplication and addition commute, the fma's product commutes) together
with their op_sel / op_sel_hi / neg_lo / neg_hi bits, which is the same arithmetic bit for bit.

    pk_opsel_fix.py in.s out.s        rewrite; prints the number of sites; exits 1 if a hazardous form is left
    pk_opsel_fix.py --check in.s      exits 1 (listing them) if the assembly contains a hazardous form
"""
import re
import sys

INSN = re.compile(r"^(\s*)(v_pk_(?:mul|add|fma)_f32)\s+(.*)$")
MOD = re.compile(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def split_operands(text):
    """'v[8:9], v[6:7], v[10:11] op_sel:[0,1]' -> (['v[8:9]', 'v[6:7]', 'v[10:11]'], {'op_sel': [0, 1]})"""
    mods = {m.group(1): [int(x) for x in m.group(2).split(",")] for m in MOD.finditer(text)}
    ops = MOD.sub("", text)
    ops = ops.split(";")[0].strip()
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out, mods


def hazardous(line):
    m = INSN.match(line)
    if not m:
        return None
    ops, mods = split_operands(m.group(3))
    sel = mods.get("op_sel")
    if not sel or sel[0] != 0 or sel[1] != 1:
        return None
    if ops[1] == ops[2]:                       # both sources the same pair: read once, never failed (form 13 of the microbenchmark)
        return None
    return m, ops, mods


def fix_line(line):
    h = hazardous(line)
    if not h:
        return line, 0
    m, ops, mods = h
    n = len(ops) - 1                           # number of sources
    ops[1], ops[2] = ops[2], ops[1]
    if "op_sel_hi" not in mods:
        mods["op_sel_hi"] = [1] * n           # the default, spelled out before it is permuted
    text = ", ".join(ops)
    for name in ("op_sel", "op_sel_hi", "neg_lo", "neg_hi"):
        if name in mods:
            v = mods[name]
            v[0], v[1] = v[1], v[0]
            if name == "op_sel_hi" and all(x == 1 for x in v):
                continue
            if name in ("op_sel", "neg_lo", "neg_hi") and not any(v):
                continue
            text += f" {name}:[{','.join(str(x) for x in v)}]"
    return f"{m.group(1)}{m.group(2)} {text}", 1


def main():
    if sys.argv[1] == "--check":
        bad = [(i + 1, l.strip()) for i, l in enumerate(open(sys.argv[2])) if hazardous(l)]
        for i, l in bad:
            print(f"{sys.argv[2]}:{i}: {l}")
        sys.exit(1 if bad else 0)
    src, dst = sys.argv[1], sys.argv[2]
    out, n = [], 0
    for l in open(src).read().split("\n"):
        new, k = fix_line(l)
        out.append(new)
        n += k
    left = [l for l in out if hazardous(l)]
    open(dst, "w").write("\n".join(out))
    print(f"pk_opsel_fix: {n} packed-fp32 instructions commuted in {src}")
    sys.exit(1 if left else 0)


if __name__ == "__main__":
    main()
