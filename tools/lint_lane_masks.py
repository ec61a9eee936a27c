#!/usr/bin/env python3
"""ISA lint for the lane-mask hazard family of profiles/experiments/r02_lane_mask.md and r02_square_branch.md (cause unconfirmed:
DESIGN.md 6.1).  Compiles each .hip file for gfx950 to assembly (compile-only, no GPU) and counts per kernel two shapes, both "a
64-bit lane mask written by a VALU compare is consumed by the scalar unit":

  S1  select on a scalar-combined mask:   v_cmp* -> s[a:b] / vcc ;  s_{and,or,andn2,orn2,xor,xnor,nand,nor}_b64 D, .., s[a:b] ;
      v_cndmask_b32 (e32 reading vcc, or e64 reading D) within WINDOW instructions of the scalar op.
      This is the shape that zeroed tap weights on lanes 48-63 in bwd_scatter_sorted_kernel (r02_lane_mask.md).  The e64 form
      with an SGPR-pair destination is included (the first lint only saw `vcc` + e32).
  S2  uniform branch on a VALU mask:      v_cmp* -> X ;  s_and_b64 vcc, exec, X ;  s_cbranch_vccz / vccnz.
      The lowering of every branch whose condition the compiler proved uniform but computed on the VALU; the shape of the
      SQUARE finding (r02_square_branch.md).  It is everywhere in compiled code (loop back-edges on VGPR-derived trip counts),
      so it is REPORTED, not forbidden: the guard for it is the repeated-launch / occupancy hash tests of the -m gpu suite.

A mask counts as "VALU-written" when the producing v_cmp is in the same basic block (labels and branches end a block), which is
where the scalar consumer can be close enough in time to matter.  S2far is the same branch shape with the v_cmp anywhere earlier
in the kernel's text (linear scan, the pair not overwritten by a scalar instruction since): masks of loop-invariant conditions.

    python tools/lint_lane_masks.py [--report FILE] [--enforce-s1 name.hip ...] [files...]
Exit code 1 when a file named by --enforce-s1 has S1 > 0 (default: all four .hip files).  tests/test_lint_cpu.py runs it
(compile-only, no GPU).
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerffaceediting_amd", "csrc")
ALL = ["nfe_render.hip", "nfe_render_bwd.hip", "nfe_planes.hip", "nfe_dense.hip"]
WINDOW = 6
SLOGIC = re.compile(r"^\s*s_(and|or|andn2|orn2|xor|xnor|nand|nor)_b64\s+(vcc|s\[\d+:\d+\]),\s*([^,]+),\s*([^\s;]+)")
VCMP = re.compile(r"^\s*v_cmp[a-z_]*_[a-z]+\d+(_e64|_e32)?\s+(vcc|s\[\d+:\d+\])?")
CNDMASK = re.compile(r"^\s*v_cndmask_b32(_e32|_e64|_dpp|_sdwa)?\s+(.*)")
BRANCH = re.compile(r"^\s*s_cbranch_vcc(z|nz)\b")
BLOCK_END = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc|s_swappc)|^[.\w$]+:")


def assemble(src, outdir, extra=()):
    out = os.path.join(outdir, os.path.basename(src) + ".s")
    deps = [src, os.path.join(CSRC, "nfe_common.h")]
    if os.path.exists(out) and not extra and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, *extra, "-x", "hip", src, "--cuda-device-only", "-S", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise SystemExit(2)
    return out


def vcmp_dest(line):
    m = re.match(r"^\s*(v_cmpx?_[a-z0-9_]+?)(_e64|_e32)?\s+(.*)", line)
    if not m or m.group(1).startswith("v_cmpx"):
        return None
    ops = m.group(3)
    first = ops.split(",")[0].strip()
    if first == "vcc" or re.match(r"s\[\d+:\d+\]$", first):
        return first
    return "vcc" if m.group(2) != "_e64" else None        # e32 compares write vcc implicitly


def scan(path):
    s1, s2, s2far = collections.Counter(), collections.Counter(), collections.Counter()
    kernel = None
    far = set()                   # SGPR pairs written by a v_cmp anywhere earlier in this kernel, not overwritten by SALU since
    and_exec_far = 0
    valu_masks = set()            # SGPR pairs / vcc written by a v_cmp in the current basic block
    pending = []                  # (dest, instructions left) of scalar logic ops on VALU-written masks
    and_exec = 0                  # instructions left in which s_cbranch_vcc* would complete an S2
    for raw in open(path):
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            far.clear()
        if line.lstrip().startswith("."):
            continue
        d = vcmp_dest(line)
        if d:
            valu_masks.add(d)
            far.add(d)
        ms = re.match(r"^\s*s_\w+\s+(vcc|s\[\d+:\d+\])\s*,", line)
        m = SLOGIC.match(line)
        if m:
            dest, a, b = m.group(2), m.group(3).strip(), m.group(4).strip()
            srcs = {a, b}
            if srcs & valu_masks:
                pending.append([dest, WINDOW + 1])
                if dest == "vcc" and m.group(1) == "and" and "exec" in srcs:
                    and_exec = 4
            if dest == "vcc" and m.group(1) == "and" and "exec" in srcs and (srcs & far):
                and_exec_far = 4
            valu_masks.discard(dest)          # overwritten by the scalar unit
        if ms:
            far.discard(ms.group(1))
        c = CNDMASK.match(line)
        if c:
            form, ops = c.group(1) or "", c.group(2)
            used = None
            if form == "_e64":
                last = ops.split(",")[-1].strip()
                used = last if (last == "vcc" or last.startswith("s[")) else None
            elif "vcc" in ops.split(",")[-1] or form in ("_e32", ""):
                used = "vcc"
            for p in pending:
                if used == p[0] and p[1] > 0:
                    s1[kernel] += 1
                    p[1] = 0
        if BRANCH.match(line) and and_exec > 0:
            s2[kernel] += 1
        if BRANCH.match(line) and and_exec_far > 0:
            s2far[kernel] += 1
        and_exec_far = max(0, and_exec_far - 1)
        if BLOCK_END.match(line):
            valu_masks.clear()
            pending = []
            and_exec = 0
        else:
            for p in pending:
                p[1] -= 1
            pending = [p for p in pending if p[1] > 0]
            and_exec = max(0, and_exec - 1)
    return s1, s2, s2far


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*\)$", "", d).replace("void ", "").replace("nfe::", "") for d in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--report")
    ap.add_argument("--enforce-s1", nargs="*", default=list(ALL))
    ap.add_argument("--asm-dir", default=os.path.join(CSRC, "build", "lint"))
    ap.add_argument("--flag", action="append", default=[], help="extra hipcc flag (e.g. -DNFE_BWD_FIX=0)")
    args = ap.parse_args()
    files = args.files or [os.path.join(CSRC, f) for f in ALL]
    os.makedirs(args.asm_dir, exist_ok=True)
    lines, rc = [], 0
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:                      # hipcc -S of the four files side by side
        asms = list(ex.map(lambda f: assemble(f, args.asm_dir, args.flag), files))
    for f, asm in zip(files, asms):
        s1, s2, s2far = scan(asm)
        kernels = sorted(set(s1) | set(s2) | set(s2far))
        t1, t2, t3 = sum(s1.values()), sum(s2.values()), sum(s2far.values())
        lines.append(f"{os.path.basename(f)}: S1 (select on scalar-combined VALU mask) = {t1}, S2 (uniform branch on VALU mask, same block) = {t2}, "
                     f"S2far (v_cmp anywhere earlier) = {t3}")
        for k, d in zip(kernels, demangle(kernels)):
            lines.append(f"    S1 {s1[k]:4d}  S2 {s2[k]:4d}  S2far {s2far[k]:4d}  {d[:110]}")
        if os.path.basename(f) in args.enforce_s1 and t1:
            rc = 1
    text = "\n".join(lines)
    print(text)
    if args.report:
        with open(args.report, "w") as fh:
            fh.write("# tools/lint_lane_masks.py: per-kernel counts of the two lane-mask shapes (see the script header)\n" + text + "\n")
    sys.exit(rc)


if __name__ == "__main__":
    main()
