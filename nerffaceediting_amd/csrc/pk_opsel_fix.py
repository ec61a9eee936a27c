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

The pass FAILS CLOSED (round 5): every `v_pk_{mul,add,fma}_f32` line is parsed completely - destination, the exact number of
sources the opcode takes, every source against the operand grammar below, every trailing modifier against the known set with the
right number of bits - and a line that does not parse is an error of the build (exit 2, the line is printed), never "not
hazardous".  A new ROCm that spells a modifier differently therefore stops the build instead of silently shipping the form.

    pk_opsel_fix.py in.s out.s        rewrite; prints the number of sites; exit 1 if a hazardous form is left, 2 on a parse failure
    pk_opsel_fix.py --check in.s      exit 1 (listing them) if the assembly / disassembly contains a hazardous form, 2 on a parse failure
    pk_opsel_fix.py --check-lib x.so  the same for the gfx950 code objects embedded in a built library (llvm-objdump -d of every
                                      bundle entry): covers ANY build path, not only the Makefile's
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OPCODES = {"v_pk_mul_f32": 2, "v_pk_add_f32": 2, "v_pk_fma_f32": 3}
INSN = re.compile(r"^(\s*)(v_pk_(?:mul|add|fma)_f32)\b(.*)$")
BIT_MODS = ("op_sel", "op_sel_hi", "neg_lo", "neg_hi")
MOD = re.compile(r"^(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01](?:,[01])*)\]$")
FLAG_MODS = ("clamp",)
# source / destination operand grammar of packed fp32 (64-bit register pairs, or constants that the hardware broadcasts)
REG_PAIR = re.compile(r"^(?:v|s|a|ttmp)\[\d+:\d+\]$")
SPECIAL = re.compile(r"^(?:vcc|exec|src_\w+|null)$")
NUMBER = re.compile(r"^-?(?:0x[0-9a-fA-F]+|\d+(?:\.\d*)?(?:[eE][-+]?\d+)?|\.\d+)$")
LIT = re.compile(r"^lit\(-?(?:0x[0-9a-fA-F]+|\d+(?:\.\d*)?)\)$")


class ParseError(Exception):
    pass


def strip_comment(text):
    """hipcc -S comments start with ';', llvm-objdump's encoding column with '//'."""
    for mark in (";", "//"):
        i = text.find(mark)
        if i >= 0:
            text = text[:i]
    return text.rstrip()


def parse(line):
    """None for a line that is not one of the three opcodes, else (indent, opcode, [dst, src0, ...], {modifier: bits}, [flags]).
    Raises ParseError for anything about such a line that is not understood."""
    m = INSN.match(line)
    if not m:
        first = line.split(None, 1)[0] if line.strip() else ""
        if first.startswith("v_pk_") and "_f32" in first:       # v_pk_max_f32, v_pk_mul_f32_dpp, ...: never seen, never tested
            raise ParseError(f"unknown packed-fp32 opcode {first!r}: teach pk_opsel_fix.py whether its operand forms are safe")
        return None
    indent, opcode, rest = m.group(1), m.group(2), strip_comment(m.group(3))
    # operands: comma separated at bracket depth 0; the LAST operand ends at the first whitespace outside brackets
    ops, depth, cur, i, rest = [], 0, "", 0, rest.strip()
    n_ops = OPCODES[opcode] + 1
    while i < len(rest):
        ch = rest[i]
        if ch == "[" or ch == "(":
            depth += 1
        elif ch == "]" or ch == ")":
            depth -= 1
            if depth < 0:
                raise ParseError("unbalanced brackets")
        if depth == 0 and ch == ",":
            ops.append(cur.strip()); cur = ""
        elif depth == 0 and ch.isspace() and len(ops) == n_ops - 1 and cur.strip():
            break
        else:
            cur += ch
        i += 1
    ops.append(cur.strip())
    tail = rest[i:].split()
    if depth != 0:
        raise ParseError("unbalanced brackets")
    if len(ops) != n_ops or not all(ops):
        raise ParseError(f"{opcode} takes {n_ops} operands, found {len(ops)}: {ops}")
    if not REG_PAIR.match(ops[0]):
        raise ParseError(f"destination {ops[0]!r} is not a register pair")
    for o in ops[1:]:
        if not (REG_PAIR.match(o) or SPECIAL.match(o) or NUMBER.match(o) or LIT.match(o)):
            raise ParseError(f"source operand {o!r} not understood")
    mods, flags = {}, []
    for t in tail:
        mm = MOD.match(t)
        if mm:
            if mm.group(1) in mods:
                raise ParseError(f"modifier {mm.group(1)} given twice")
            bits = [int(x) for x in mm.group(2).split(",")]
            if len(bits) != n_ops - 1:
                raise ParseError(f"{mm.group(1)} has {len(bits)} bits for {n_ops - 1} sources")
            mods[mm.group(1)] = bits
        elif t in FLAG_MODS:
            flags.append(t)
        else:
            raise ParseError(f"modifier {t!r} not understood")
    return indent, opcode, ops, mods, flags


def hazardous(line):
    p = parse(line)
    if p is None:
        return None
    indent, opcode, ops, mods, flags = p
    sel = mods.get("op_sel")
    if not sel or sel[0] != 0 or sel[1] != 1:
        return None
    if ops[1] == ops[2]:                       # both sources the same pair: read once, never failed (form 13 of the microbenchmark)
        return None
    return p


def fix_line(line):
    h = hazardous(line)
    if not h:
        return line, 0
    indent, opcode, ops, mods, flags = h
    n = len(ops) - 1                           # number of sources
    ops[1], ops[2] = ops[2], ops[1]
    if "op_sel_hi" not in mods:
        mods["op_sel_hi"] = [1] * n           # the default, spelled out before it is permuted
    text = ", ".join(ops)
    for name in BIT_MODS:
        if name in mods:
            v = mods[name]
            v[0], v[1] = v[1], v[0]
            if name == "op_sel_hi" and all(x == 1 for x in v):
                continue
            if name in ("op_sel", "neg_lo", "neg_hi") and not any(v):
                continue
            text += f" {name}:[{','.join(str(x) for x in v)}]"
    for f in flags:
        text += " " + f
    return f"{indent}{opcode} {text}", 1


def scan(path_or_lines, label):
    """-> (hazardous lines, parse failures), each as (location, text)."""
    lines = open(path_or_lines).read().split("\n") if isinstance(path_or_lines, str) else path_or_lines
    bad, broken = [], []
    for i, l in enumerate(lines):
        try:
            if hazardous(l):
                bad.append((f"{label}:{i + 1}", l.strip()))
        except ParseError as e:
            broken.append((f"{label}:{i + 1}", f"{l.strip()}    <- {e}"))
    return bad, broken


BUNDLE_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin")


def device_code_objects(lib_path, arch="gfx950"):
    """The code objects of `arch` inside a host library's offload bundles (clang-offload-bundler format: magic, u64 count, then per
    entry u64 offset, u64 size, u64 triple length, triple)."""
    blob = open(lib_path, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(BUNDLE_MAGIC, pos)
        if pos < 0:
            break
        p = pos + len(BUNDLE_MAGIC)
        (count,) = struct.unpack_from("<Q", blob, p); p += 8
        if count > 64:
            pos += 1
            continue
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p); p += 24
            triple = blob[p:p + tlen].decode(errors="replace"); p += tlen
            if arch in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = p
    return out


def check_library(lib_path):
    objs = device_code_objects(lib_path)
    if not objs:
        print(f"pk_opsel_fix: no gfx950 code object found in {lib_path}")
        return 2
    bad, broken, n_pk = [], [], 0
    for k, obj in enumerate(objs):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj); f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True)
        if dis.returncode != 0:
            print(f"pk_opsel_fix: llvm-objdump failed on code object {k} of {lib_path}: {dis.stderr[-300:]}")
            return 2
        lines = dis.stdout.split("\n")
        n_pk += sum(1 for l in lines if INSN.match(l))
        b, br = scan(lines, f"{os.path.basename(lib_path)}[{k}]")
        bad += b; broken += br
    for loc, l in broken:
        print(f"PARSE FAILURE {loc}: {l}")
    for loc, l in bad:
        print(f"HAZARD {loc}: {l}")
    print(f"pk_opsel_fix: {len(objs)} code objects, {n_pk} packed-fp32 mul/add/fma instructions, {len(bad)} hazardous, {len(broken)} unparsed in {lib_path}")
    return 2 if broken else (1 if bad else 0)


def main():
    if sys.argv[1] == "--check-lib":
        sys.exit(check_library(sys.argv[2]))
    if sys.argv[1] == "--check":
        bad, broken = scan(sys.argv[2], sys.argv[2])
        for loc, l in broken:
            print(f"PARSE FAILURE {loc}: {l}")
        for loc, l in bad:
            print(f"{loc}: {l}")
        sys.exit(2 if broken else (1 if bad else 0))
    src, dst = sys.argv[1], sys.argv[2]
    out, n = [], 0
    for i, l in enumerate(open(src).read().split("\n")):
        try:
            new, k = fix_line(l)
        except ParseError as e:
            print(f"pk_opsel_fix: PARSE FAILURE {src}:{i + 1}: {l.strip()}    <- {e}")
            print("pk_opsel_fix: refusing to build: a packed-fp32 instruction this pass cannot read might be the hazardous form")
            sys.exit(2)
        out.append(new)
        n += k
    left, broken = scan(out, dst)
    open(dst, "w").write("\n".join(out))
    print(f"pk_opsel_fix: {n} packed-fp32 instructions commuted in {src}")
    sys.exit(2 if broken else (1 if left else 0))


if __name__ == "__main__":
    main()
