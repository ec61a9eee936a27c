#!/usr/bin/env python3
"""Per-opcode census of the innermost loop around a kernel's MFMA instructions (no line tables needed: reads csrc/build/*.hip.s).

    python3 tools/opcode_census.py nerffaceediting_amd/csrc/build/nfe_render.hip.s <mangled-kernel-substring> [--json out.json]
                                   [--anchor global_load_dwordx4]      (the producer loop of a wave-specialised kernel)
                                   [--loops [--pick <first instruction>]]  list every loop that holds anchors / census of one of them

Finer than isa_census.py's issue classes: what the vector instructions outside the transcendental / packed / MFMA classes ARE
(moves, selects, byte permutes, bit operations, conversions, compares, min / max, plain fp32 arithmetic, lane reads).
"""
import json
import re
import sys
from collections import Counter

GROUPS = [
    ("mfma", lambda m: m.startswith("v_mfma")),
    ("trans (exp/log/rcp/rsq/sqrt)", lambda m: m.startswith(("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_"))),
    ("packed fp32 (v_pk_*)", lambda m: m.startswith("v_pk_")),
    ("move (v_mov / v_accvgpr)", lambda m: m.startswith(("v_mov_", "v_accvgpr"))),
    ("select (v_cndmask)", lambda m: m.startswith("v_cndmask")),
    ("byte permute (v_perm_b32)", lambda m: m.startswith("v_perm_b32")),
    ("bit ops (and/or/xor/shift/bfe/bfi)", lambda m: m.startswith(("v_and_", "v_or_", "v_xor_", "v_lshl", "v_lshr", "v_ashr", "v_bfe_", "v_bfi_", "v_not_", "v_and_or", "v_or3", "v_lshl_or", "v_lshl_add"))),
    ("convert (v_cvt_*)", lambda m: m.startswith("v_cvt_")),
    ("compare (v_cmp*)", lambda m: m.startswith("v_cmp")),
    ("min / max / med3", lambda m: m.startswith(("v_min", "v_max", "v_med3"))),
    ("fp32 fma / mul / add / sub", lambda m: m.startswith(("v_fma_", "v_fmac_", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mad_", "v_mul_legacy", "v_ldexp", "v_frexp", "v_fract", "v_floor", "v_rndne", "v_trunc"))),
    ("integer add / mul", lambda m: m.startswith(("v_add_u32", "v_add_co", "v_addc", "v_sub_u32", "v_sub_co", "v_subrev_u32", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_add3", "v_mad_u", "v_mad_i", "v_add_lshl", "v_sub_i", "v_add_i", "v_subrev_co"))),
    ("lane ops (readlane / permlane / dpp / swizzle)", lambda m: m.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")) or "dpp" in m),
    ("LDS", lambda m: m.startswith("ds_")),
    ("vector memory", lambda m: m.startswith(("global_", "buffer_", "flat_", "scratch_"))),
    ("waits / nops / sleep", lambda m: m.startswith(("s_waitcnt", "s_nop", "s_sleep"))),
    ("scalar", lambda m: m.startswith("s_")),
]


def main():
    path, kern = sys.argv[1], sys.argv[2]
    text = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(text) if re.match(r"^_Z\S*%s\S*:" % re.escape(kern), ln))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    insts, labels = [], {}
    for i in range(start + 1, end):
        ln = text[i].split(";")[0].rstrip()
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", ln)
        if m and not m.group(1).startswith("."):
            mn = m.group(1) + ("_dpp" if ("quad_perm" in m.group(2) or "row_" in m.group(2)) and "dpp" not in m.group(1) else "")
            insts.append((mn, m.group(2)))
    anchor = sys.argv[sys.argv.index("--anchor") + 1] if "--anchor" in sys.argv else "v_mfma"      # the loop is found around these
    mf = [k for k, it in enumerate(insts) if it[0].startswith(anchor)]
    best = None
    for k, it in enumerate(insts):
        if it[0].startswith(("s_cbranch", "s_branch")):
            tgt = labels.get(it[1].strip())
            if tgt is not None and tgt < k:
                inside = sum(1 for a in mf if tgt <= a <= k)
                if 2 * inside >= len(mf) and (best is None or (k - tgt) < (best[1] - best[0])):
                    best = (tgt, k)
    if "--loops" in sys.argv or best is None:          # several loops share the anchors (template variants inlined side by side): list them
        loops = []
        for k, it in enumerate(insts):
            if it[0].startswith(("s_cbranch", "s_branch")):
                tgt = labels.get(it[1].strip())
                if tgt is not None and tgt < k:
                    loops.append((sum(1 for a in mf if tgt <= a <= k), k - tgt + 1, tgt, k))
        loops = [lp for lp in loops if lp[0]]
        for n_anchor, size, tgt, k in sorted(loops, key=lambda t: t[2]):
            print("loop at instructions %6d..%6d: %5d instructions, %4d anchors" % (tgt, k, size, n_anchor))
        pick = int(sys.argv[sys.argv.index("--pick") + 1]) if "--pick" in sys.argv else None
        if pick is None:
            return
        best = next((tgt, k) for n_anchor, size, tgt, k in loops if tgt == pick)
    body = insts[best[0]:best[1] + 1]
    ops = Counter(mn for mn, _ in body)
    groups = Counter()
    members = {}
    for mn, n in ops.items():
        g = next((name for name, f in GROUPS if f(mn)), "other vector" if mn.startswith("v_") else "other")
        groups[g] += n
        members.setdefault(g, Counter())[mn] += n
    valu = sum(n for mn, n in ops.items() if mn.startswith("v_") and not mn.startswith("v_mfma"))
    print("kernel %s: loop body %d instructions, %d vector ALU (without MFMA), %d MFMA" % (kern, len(body), valu, groups["mfma"]))
    for g, n in groups.most_common():
        print("%5d  %-48s %s" % (n, g, ", ".join("%s %d" % kv for kv in members[g].most_common(6))))
    if "--json" in sys.argv:
        json.dump({"kernel": kern, "loop_instructions": len(body), "valu": valu, "groups": dict(groups),
                   "opcodes": dict(ops)}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
