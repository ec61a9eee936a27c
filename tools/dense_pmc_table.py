#!/usr/bin/env python3
"""profiles/rNN_pmc_dense_<math>.txt (tools/pmc_summary.py output of a `tools/time_full.py` run under tools/pmc.sh) -> one JSON table of
the conv kernels' counter-derived figures, which bench.py prints in its dense roofline blocks (`kernels`).

    dense_pmc_table.py OUT.json math=summary.txt [math=summary.txt ...]         e.g.  bf16=profiles/r05_pmc_dense_bf16.txt

Per kernel variant (every conv3_kernel / upfir / torgb kernel that takes >= 2 % of the run's GPU time):
    matrix_pipe      SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
    lds_bank_conflict_per_launch, share_of_time (of all kernels in the summary), avg_us, dispatches
"""
import json
import re
import sys


def parse(path):
    kernels, cur = {}, None
    for line in open(path):
        m = re.match(r"^== (.*?)\s+dispatches/pass~(\d+)\s+avg_ns=(\d+)", line)
        if m:
            cur = {"dispatches": int(m.group(2)), "avg_ns": float(m.group(3))}
            kernels[m.group(1).strip()] = cur
            continue
        m = re.match(r"^\s+(\w+)\s+avg/dispatch\s+([0-9.e+\-]+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return kernels


def table(path):
    ks = parse(path)
    total = sum(k["dispatches"] * k["avg_ns"] for k in ks.values()) or 1.0
    out = {"source": path}
    for name, k in sorted(ks.items(), key=lambda kv: -kv[1]["dispatches"] * kv[1]["avg_ns"]):
        share = k["dispatches"] * k["avg_ns"] / total
        if share < 0.02 or "GRBM_GUI_ACTIVE" not in k:
            continue
        cycles = k["GRBM_GUI_ACTIVE"] / 8.0
        short = re.sub(r"^void nfe::|\(nfe::\w+\)$", "", name)
        e = {"share_of_time": round(share, 4), "avg_us": round(k["avg_ns"] / 1e3, 2), "dispatches": k["dispatches"]}
        if k.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["matrix_pipe"] = round(k["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles), 4)
        if "SQ_LDS_BANK_CONFLICT" in k:
            e["lds_bank_conflict_per_launch"] = k["SQ_LDS_BANK_CONFLICT"]
        if k.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None and k.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            e["mfma_coexec_share"] = round(k["SQ_VALU_MFMA_COEXEC_CYCLES"] / k["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
        out[short] = e
    return out


def main():
    out = {}
    for arg in sys.argv[2:]:
        math, path = arg.split("=", 1)
        out[math] = table(path)
    json.dump(out, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
