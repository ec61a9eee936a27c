#!/usr/bin/env python3
"""Audit of every instruction-bearing inline-asm statement of the library against the hazards hipcc does NOT handle for `asm`
statements (cdna_hip_programming.md 5.7: nothing inside the string is padded, memory operations are not counted, M0 is not
preserved; round 3 found one such case the hard way: a stale M0 after s_set_gpr_idx_on, profiles/experiments/r03_lane_mask.md).

Compile-only (no GPU).  For each of the four .hip files the ISA is scanned between `;;#ASMSTART` / `;;#ASMEND`; statements that
contain no instruction (the empty `asm volatile("" : "+v"(x))` optimisation barriers, comments) are counted and skipped.  For
every instruction-bearing block, per kernel instance, the rules below are CHECKED on the emitted ISA:

  M0-1  SALU write of M0 (s_mov_b32 m0 / s_set_gpr_idx_on / s_set_gpr_idx_idx) -> first consumer of M0 (LDS-DMA, indexed VALU)
        INSIDE the same statement, with wait states between them: >= 1 for LDS-DMA (the guide's recipe: s_nop 0), >= 4 for the
        VGPR index mode (s_nop 3: the load-bearing value bisected in round 3; no published number was available offline).
  M0-2  kernels that use the VGPR index mode: no instruction outside asm statements reads or writes M0, and no s_set_gpr_idx_*
        outside them (the compiler keeps nothing in M0 that the statements could clobber, and vice versa).
  M0-3  index mode is switched off (s_set_gpr_idx_off) before the statement ends.
  RSV   kernels whose asm names fixed VGPRs (the accumulate tile v[80..160]): private segment 0, no VGPR/SGPR spills, the
        register count covers the tile (161), and no compiler instruction outside asm touches v80..v160.
  SGPR  a VMEM instruction inside a statement that takes an SGPR base/offset: no VALU write (v_readlane / v_readfirstlane /
        v_cmp) of that SGPR within the 5 preceding instructions (VALU-writes-SGPR -> VMEM-reads: 5 wait states).
  MFMA  no statement contains or directly follows an MFMA whose result it reads (none of the statements takes an MFMA result
        as an operand: checked as "no v_mfma within 12 instructions before a block that reads one of its D registers").
  LAUN  the "launder" statements (`asm volatile("; nfe_launder %0" : "+v"(x))`: no instruction, they only hide a value from the
        optimiser).  For hipcc's hazard recogniser the STATEMENT is now the writer of x, so it pads nothing between the value's real
        producer and its consumers.  Checked on the ISA around every such statement (the comment names the register): the real
        producer within the 4 preceding instructions and the consumers within the 4 following ones must not form a pair that needs
        wait states on gfx940-class hardware - transcendental -> non-transcendental VALU (1), VALU -> DPP / v_readlane /
        v_readfirstlane / v_permlane (2 / 1), MFMA -> anything (many), VALU -> MFMA A/B operand (2), VALU-written SGPR -> VMEM (5) /
        lane select (4) - unless that many states separate them.  (Found the hard way in round 4: an `asm("v_add_f32 ...")` that
        consumed a v_exp_f32 result gave run-dependent results, profiles/experiments/r04_asm_trans_hazard.md.)
  TRNS  a VALU instruction INSIDE a statement that reads a register a transcendental (v_exp / v_log / v_rcp / ...) wrote in the
        instruction right before it, with no wait state between (gfx940 "trans forwarding" hazard: hipcc pads it for its own
        instructions only).  Same check for MFMA results read inside a statement (rule MFMA).
  WAIT  statements that only wait (s_waitcnt) are listed; they order the COMPILER's counted loads (LDS-DMA is issued by asm and
        counted by hand: conv3_kernel's vmcnt ladder).

usage: asm_audit.py [--md OUT.md] [--files a.hip,b.hip]     exit code 1 on any violated rule
       ASM_AUDIT_FLAGS="-DNFE_SOFTPLUS_SCALAR=1" asm_audit.py --files nfe_render.hip     audits an experiment build (own ISA directory)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import lint_lane_masks as L  # noqa: E402  (assemble(): the same compile as the lane-mask lint)

INST = re.compile(r"^\s*([a-z_][a-z_0-9]*)\b\s*(.*?)\s*(?:;.*)?$")


def kernels(text):
    """yield (name, lines, meta) for every kernel of an assembly file"""
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):\s*; @", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        meta = {}
        k = j
        while k < len(lines) and k < j + 80:
            mm = re.match(r"^\s*\.set\s+%s\.(\w+),\s*(\S+)" % re.escape(name), lines[k])
            if mm:
                meta[mm.group(1)] = mm.group(2)
            mm = re.match(r"^;\s*(ScratchSize|NumVgprs|NumSgprs|Occupancy|LDSByteSize|SGPRSpill|VGPRSpill|codeLenInByte):\s*(\S+)", lines[k].replace("; ", ";", 1)) \
                or re.match(r"^; (ScratchSize|NumVgprs|NumSgprs|Occupancy|LDSByteSize|SGPRSpill|VGPRSpill|codeLenInByte): (\S+)", lines[k])
            if mm:
                meta[mm.group(1)] = mm.group(2)
            k += 1
        yield name, lines[i + 1:j], meta
        i = j


def parse(lines):
    """-> list of (mnemonic, operands, in_asm_block_id or None)"""
    out, blk, nblk = [], None, 0
    for ln in lines:
        if ";;#ASMSTART" in ln:
            blk = nblk
            nblk += 1
            continue
        if ";;#ASMEND" in ln:
            blk = None
            continue
        if re.match(r"^\s*\.", ln) or re.match(r"^[.\w$]+:", ln):
            continue
        m = INST.match(ln)
        if m:
            out.append((m.group(1), m.group(2), blk))
    return out


def nop_states(mn, ops):
    if mn == "s_nop":
        return int(ops.split()[0], 0) + 1
    if mn == "v_nop":
        return 1
    return 0


def uses_m0(mn, ops):
    return bool(re.search(r"\bm0\b", ops)) or mn.startswith("s_set_gpr_idx") or " lds" in (" " + ops) or mn.startswith("global_load_lds") \
        or mn.startswith(("s_movrel", "v_movrel", "ds_gws", "s_sendmsg")) or "addtid" in mn


def regs(ops, kind):
    found = set()
    for a, b in re.findall(r"\b%s\[(\d+):(\d+)\]" % kind, ops):
        found.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\b%s(\d+)\b" % kind, ops):
        found.add(int(a))
    return found


TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
LANE_X = ("v_readlane", "v_readfirstlane", "v_permlane", "v_writelane")


def launder_sites(lines):
    """-> list of (instruction index the statement sits in front of, kind 'v'/'s', set of registers) for the annotated statements"""
    out, n, in_blk = [], 0, False
    for ln in lines:
        if ";;#ASMSTART" in ln:
            in_blk = True
            continue
        if ";;#ASMEND" in ln:
            in_blk = False
            continue
        m = re.search(r";\s*nfe_launder\s+(.*)$", ln)
        if in_blk and m:
            txt = m.group(1)
            rv, rs = regs(txt, "v"), regs(txt, "s")
            if rv:
                out.append((n, "v", rv))
            if rs:
                out.append((n, "s", rs))
            continue
        if re.match(r"^\s*\.", ln) or re.match(r"^[.\w$]+:", ln):
            continue
        if INST.match(ln) and not in_blk:
            n += 1
        elif INST.match(ln) and in_blk:
            n += 1
    return out


def check_launder(name, lines, errors, stats):
    ins = [(m, o) for m, o, _ in parse(lines)]
    for at, kind, rset in launder_sites(lines):
        stats["launder"] = stats.get("launder", 0) + 1
        # the real producer: the closest earlier instruction whose destination (first operand) holds one of the registers
        prod, states = None, 0
        for k in range(at - 1, max(-1, at - 5), -1):
            mn, ops = ins[k]
            n = nop_states(mn, ops)
            if n:
                states += n
                continue
            if mn.startswith(("s_waitcnt",)):
                continue
            dst = ops.split(",")[0]
            if regs(dst, kind) & rset and not mn.startswith(("global_store", "ds_write", "buffer_store", "s_cmp", "v_cmp")):
                prod = (mn, at - 1 - k)
                break
            states += 1
        if prod is None:
            continue
        pm = prod[0]
        for k in range(at, min(len(ins), at + 4)):
            mn, ops = ins[k]
            n = nop_states(mn, ops)
            if n:
                states += n
                continue
            srcs = ",".join(ops.split(",")[1:]) if not mn.startswith(("global_store", "ds_write", "buffer_store")) else ops
            if regs(srcs, kind) & rset:
                need = 0
                if kind == "v":
                    if pm.startswith("v_mfma"):
                        need = 12
                    elif pm.startswith(TRANS) and mn.startswith("v_") and not mn.startswith(TRANS):
                        need = 1
                    elif pm.startswith("v_") and ("dpp" in mn or "quad_perm" in ops or "row_" in ops):
                        need = 2
                    elif pm.startswith("v_") and mn.startswith(LANE_X):
                        need = 1
                    elif pm.startswith("v_") and mn.startswith("v_mfma"):
                        need = 2
                else:
                    if pm.startswith("v_") and mn.startswith(("global_", "buffer_", "scratch_")):
                        need = 5
                    elif pm.startswith("v_") and mn.startswith(("v_readlane", "v_writelane")):
                        need = 4
                if need:
                    stats["launder_pairs"] = stats.get("launder_pairs", 0) + 1
                    if states < need:
                        errors.append("%s: LAUN: %s -> [launder] -> %s with %d wait states (< %d)" % (name, pm, mn, states, need))
            states += 1


def audit_kernel(name, lines, meta, report, errors):
    ins = parse(lines)
    blocks = {}
    for idx, (mn, ops, blk) in enumerate(ins):
        if blk is not None:
            blocks.setdefault(blk, []).append(idx)
    real = {b: ix for b, ix in blocks.items() if ix}
    idx_mode = False
    fixed_tile = False
    for b, ix in real.items():
        text = [(ins[i][0], ins[i][1]) for i in ix]
        sig = "; ".join(m + (" " + o if o else "") for m, o in text)
        sig = re.sub(r"v\[\d+\+[^\]]*\]", "v[tile+k]", sig)
        sig = re.sub(r"\b([vsa])\[?\d+(:\d+)?\]?", r"\1#", sig)
        entry = report.setdefault(sig, {"kernels": set(), "count": 0, "rules": set()})
        entry["kernels"].add(name)
        entry["count"] += 1
        # ---- M0-1 / M0-3
        pend, states, mode_on = None, 0, False
        for mn, ops in text:
            if mn in ("s_set_gpr_idx_on", "s_set_gpr_idx_idx") or (mn.startswith("s_") and re.match(r"^m0\b", ops)):
                pend, states = mn, 0
                if mn == "s_set_gpr_idx_on":
                    mode_on = True
                idx_mode = idx_mode or mn.startswith("s_set_gpr_idx")
                continue
            if mn == "s_set_gpr_idx_off":
                mode_on, pend = False, None
                continue
            n = nop_states(mn, ops)
            if n:
                states += n
                continue
            if pend is not None:
                need = None
                if pend.startswith("s_set_gpr_idx") and mn.startswith("v_"):
                    need = 4
                elif "lds" in mn or " lds" in (" " + ops):
                    need = 1
                if need is not None:
                    entry["rules"].add("M0-1 (%d >= %d states)" % (states, need))
                    if states < need:
                        errors.append("%s: M0-1: %s -> %s with %d wait states (< %d)" % (name, pend, mn, states, need))
                    pend = None
        if mode_on:
            errors.append("%s: M0-3: statement ends with the VGPR index mode still on" % name)
        if any(m.startswith("s_set_gpr_idx") for m, _ in text):
            entry["rules"].add("M0-3")
        # ---- SGPR: VALU-written SGPR -> VMEM in the statement
        for i in ix:
            mn, ops, _ = ins[i]
            if mn.startswith(("global_", "buffer_", "scratch_")):
                ss = regs(ops, "s")
                for k in range(max(0, i - 5), i):
                    pm, po, _ = ins[k]
                    if pm.startswith(("v_readlane", "v_readfirstlane", "v_cmp")) and regs(po.split(",")[0], "s") & ss:
                        errors.append("%s: SGPR: %s writes an SGPR that %s reads %d instructions later" % (name, pm, mn, i - k))
                entry["rules"].add("SGPR")
        # ---- MFMA: result consumed by the statement
        first = ix[0]
        reads = set()
        for i in ix:
            reads |= regs(",".join(ins[i][1].split(",")[1:]), "v")
        for k in range(max(0, first - 12), first):
            pm, po, _ = ins[k]
            if pm.startswith("v_mfma") and regs(po.split(",")[0], "v") & reads:
                errors.append("%s: MFMA: a statement reads the result of %s %d instructions after it" % (name, pm, first - k))
        # ---- TRNS: transcendental result consumed by a VALU instruction of the statement without a wait state
        for i in ix:
            mn, ops, _ = ins[i]
            if not mn.startswith("v_") or mn.startswith(TRANS):
                continue
            srcs = regs(",".join(ops.split(",")[1:]), "v")
            states = 0
            for k in range(i - 1, max(-1, i - 3), -1):
                pm, po, _ = ins[k]
                n = nop_states(pm, po)
                if n:
                    states += n
                    continue
                if pm.startswith(TRANS) and regs(po.split(",")[0], "v") & srcs and states < 1:
                    errors.append("%s: TRNS: %s reads the result of %s %d instruction(s) earlier with %d wait states" % (name, mn, pm, i - k, states))
                    entry["rules"].add("TRNS")
                states += 1
        if any(re.search(r"v\[80\+|\bv(8\d|9\d|1[0-5]\d|160)\b", ins[i][1]) and ins[i][0].startswith("v_") and
               re.search(r"nfe_i|\+", ins[i][1]) for i in ix):
            fixed_tile = True
    if idx_mode:
        for mn, ops, blk in ins:
            if blk is None and (re.search(r"\bm0\b", ops) or mn.startswith("s_set_gpr_idx")):
                errors.append("%s: M0-2: `%s %s` outside the asm statements of an index-mode kernel" % (name, mn, ops))
    if idx_mode or fixed_tile:
        scratch = int(meta.get("private_seg_size", meta.get("ScratchSize", "0")) or 0)
        nv = int(meta.get("num_vgpr", meta.get("NumVgprs", "0")) or 0)
        if scratch != 0:
            errors.append("%s: RSV: private segment %d bytes (a spill could land in the reserved registers)" % (name, scratch))
        if nv != 161:
            errors.append("%s: RSV: num_vgpr %d, expected 161 (tile v80..v160)" % (name, nv))
        for mn, ops, blk in ins:
            if blk is None and mn.startswith(("v_", "ds_", "global_", "buffer_")) and any(80 <= r <= 160 for r in regs(ops, "v")):
                errors.append("%s: RSV: compiler instruction `%s %s` touches the reserved tile registers" % (name, mn, ops))
                break
    return len(blocks), len(real)


def main():
    md = sys.argv[sys.argv.index("--md") + 1] if "--md" in sys.argv else None
    outdir = os.path.join(L.ROOT, "nerffaceediting_amd", "csrc", "build", "lint")
    os.makedirs(outdir, exist_ok=True)
    errors, rows = [], []
    extra = tuple(os.environ.get("ASM_AUDIT_FLAGS", "").split())          # e.g. -DNFE_SOFTPLUS_SCALAR=1: audit an experiment build
    if extra:                                                             # ... into its own directory: build/lint holds the shipped build's ISA
        outdir = os.path.join(L.ROOT, "nerffaceediting_amd", "csrc", "build", "lint_experiment")
        os.makedirs(outdir, exist_ok=True)
    files = sys.argv[sys.argv.index("--files") + 1].split(",") if "--files" in sys.argv else L.ALL
    for f in files:
        text = open(L.assemble(os.path.join(L.CSRC, f), outdir, extra)).read()
        report, nk, nb, nr = {}, 0, 0, 0
        stats = {}
        for name, lines, meta in kernels(text):
            a, b = audit_kernel(name, lines, meta, report, errors)
            check_launder(name, lines, errors, stats)
            nk, nb, nr = nk + 1, nb + a, nr + b
        print("%s: %d kernels, %d asm statements in the ISA, %d carry instructions, %d distinct; %d launder statements, %d of them between a "
              "hazard-prone producer / consumer pair" % (f, nk, nb, nr, len(report), stats.get("launder", 0), stats.get("launder_pairs", 0)))
        for sig, e in sorted(report.items(), key=lambda kv: -kv[1]["count"]):
            rows.append((f, sig, e["count"], len(e["kernels"]), ", ".join(sorted(e["rules"])) or "-"))
            print("   x%-5d in %3d kernels  [%s]  %s" % (e["count"], len(e["kernels"]), ", ".join(sorted(e["rules"])) or "no hazard rule applies", sig[:150]))
    for e in errors:
        print("VIOLATION " + e)
    print("asm audit: %d violations" % len(errors))
    if md:
        with open(md, "w") as fh:
            fh.write("| file | statement (registers anonymised) | instances | kernels | rules checked |\n|---|---|---|---|---|\n")
            for f, sig, c, k, r in rows:
                fh.write("| `%s` | `%s` | %d | %d | %s |\n" % (f, sig.replace("|", "\\|")[:400], c, k, r))
            fh.write("\n%d violations\n" % len(errors))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
