#!/usr/bin/env python3
"""Instruction census of a kernel's hot loop, by source region and by issue class.

Reads assembly produced with line tables (hipcc ... -gline-tables-only -S --cuda-device-only), finds the innermost loop that
contains the kernel's MFMA instructions (the per-sample loop of render_kernel), and attributes every instruction in it to
  * a SOURCE region: the innermost inlined source line of its `.loc` directive, mapped through the REGIONS table below;
  * an ISSUE class: plain VALU, packed-fp32 VALU (shares the matrix pipe), transcendental, DPP / permute (4-cycle), MFMA, LDS,
    vector memory, scalar, waits.
The cycle column prices each class with the single-wave issue costs of tools/microbench/valu_rate.hip (profiles/r02_valu_rate.txt).

usage: isa_census.py <file.s> <mangled-kernel-substring> [--json out.json] [--anchor global_load_dwordx4]
"""
import json
import re
import sys
from collections import Counter, defaultdict

# (file suffix, first line, last line, region): line ranges of csrc/ at the commit the census was taken (profiles/r04_isa_census*.txt)
REGIONS = [
    ("nfe_common.h", 61, 75, "philox / jitter"),
    ("nfe_common.h", 98, 193, "tap geometry"),
    ("nfe_render.hip", 84, 99, "march softplus/exp"),
    ("nfe_render.hip", 101, 130, "decoder softplus"),
    ("nfe_render.hip", 132, 138, "hi/lo bf16 split"),
    ("nfe_render.hip", 140, 143, "launder"),
    ("nfe_render.hip", 154, 170, "gather: quad broadcast / load / swizzle"),
    ("nfe_render.hip", 199, 237, "per-plane affine"),
    ("nfe_render.hip", 329, 389, "gather: pipeline (offsets, loads, bilinear FMA)"),
    ("nfe_render.hip", 391, 424, "LDS exchange quad->own"),
    ("nfe_render.hip", 556, 584, "decoder bias / hidden split glue"),
    ("nfe_render.hip", 586, 661, "decoder MFMA + fragment reads"),
    ("nfe_render.hip", 663, 692, "decoder sigmoid + dispatch"),
    ("nfe_render.hip", 694, 726, "eval_point glue (shift init)"),
    ("nfe_render.hip", 874, 909, "depth schedule + sample position"),
    ("nfe_render.hip", 911, 966, "march (composite)"),
]

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
FOURCYC = ("v_perm_b32", "v_cvt_pk_bf16", "v_permlane", "v_readlane", "v_writelane", "v_readfirstlane")
COST = {"valu": 4.94, "valu_pk": 5.4, "valu_trans": 8.6, "valu_dpp/perm": 5.1, "mfma": 8.0, "lds": 4.0, "vmem": 4.0,
        "salu": 1.0, "smem": 1.0, "wait/nop": 1.0, "branch": 1.0, "other": 1.0}


def issue_class(mn, ops):
    if mn.startswith("v_mfma"):
        return "mfma"
    if mn.startswith(TRANS):
        return "valu_trans"
    if mn.startswith("v_pk_"):
        return "valu_pk"
    if "dpp" in mn or "quad_perm" in ops or "row_" in ops or mn.startswith(FOURCYC):
        return "valu_dpp/perm"
    if mn.startswith("v_"):
        return "valu"
    if mn.startswith("ds_"):
        return "lds"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith(("s_waitcnt", "s_nop", "s_sleep")):
        return "wait/nop"
    if mn.startswith(("s_cbranch", "s_branch", "s_barrier")):
        return "branch"
    if mn.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    return "other"


def region_of(fname, line):
    for suffix, lo, hi, name in REGIONS:
        if fname.endswith(suffix) and lo <= line <= hi:
            return name
    return "other (%s:%d)" % (fname.rsplit("/", 1)[-1], line // 50 * 50)


def main():
    path, kern = sys.argv[1], sys.argv[2]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    text = open(path).read().split("\n")
    files = {}
    for ln in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(2) + "/" + m.group(3)) if m.group(3) else m.group(2)
    start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\S*%s\S*:" % re.escape(kern), ln))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    insts = []          # (index, label-or-None, mnemonic, operands, file, line)
    cur = ("?", 0)
    labels = {}
    for i in range(start + 1, end):
        ln = text[i].split(";")[0].rstrip()
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", ln)
        if m and not m.group(1).startswith("."):
            insts.append((m.group(1), m.group(2), cur[0], cur[1]))
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "v_mfma"      # the loop is found around these
    mf = [k for k, it in enumerate(insts) if it[0].startswith(anchor)]
    # innermost loop around the anchors: the backward branch with the smallest span that still holds most of them (anchors may also
    # occur in the prologue - the decoder staging loads - so "all" would find no loop)
    best = None
    for k, it in enumerate(insts):
        if it[0].startswith(("s_cbranch", "s_branch")):
            tgt = labels.get(it[1].strip())
            if tgt is not None and tgt < k:
                inside = sum(1 for a in mf if tgt <= a <= k)
                if 2 * inside >= len(mf) and (best is None or (k - tgt) < (best[1] - best[0])):
                    best = (tgt, k)
    lo, hi = best
    body = insts[lo:hi + 1]
    by_region = defaultdict(Counter)
    by_class = Counter()
    for mn, ops, f, line in body:
        c = issue_class(mn, ops)
        by_class[c] += 1
        by_region[region_of(f, line)][c] += 1
    classes = ["valu", "valu_pk", "valu_trans", "valu_dpp/perm", "mfma", "lds", "vmem", "salu", "smem", "wait/nop", "branch", "other"]
    valu_cls = classes[:4]
    print("kernel %s: loop body %d instructions (asm lines %d..%d of the function)" % (kern, len(body), lo, hi))
    hdr = "%-50s" % "region" + "".join("%9s" % c[:8] for c in classes) + "%9s%9s" % ("VALU", "issue_cy")
    print(hdr)
    rows = []
    for reg, cnt in sorted(by_region.items(), key=lambda kv: -sum(kv[1][c] * COST[c] for c in classes)):
        valu = sum(cnt[c] for c in valu_cls)
        cyc = sum(cnt[c] * COST[c] for c in classes)
        rows.append({"region": reg, "valu": valu, "issue_cycles": round(cyc), **{c: cnt[c] for c in classes if cnt[c]}})
        print("%-50s" % reg[:50] + "".join("%9d" % cnt[c] for c in classes) + "%9d%9d" % (valu, cyc))
    tot_valu = sum(by_class[c] for c in valu_cls)
    tot_cyc = sum(by_class[c] * COST[c] for c in classes)
    print("%-50s" % "TOTAL" + "".join("%9d" % by_class[c] for c in classes) + "%9d%9d" % (tot_valu, tot_cyc))
    if out_json:
        json.dump({"kernel": kern, "loop_instructions": len(body), "by_class": dict(by_class), "valu_total": tot_valu,
                   "issue_cycles_single_wave": round(tot_cyc), "cost_model": COST, "regions": rows}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
