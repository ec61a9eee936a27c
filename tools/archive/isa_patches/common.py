"""Helpers for tools/isa_patch_build.sh patchers: edit only bwd_scatter_sorted_kernel<MFMA=true, BINNED=true> of nfe_render_bwd.hip."""
import re
import sys

KERNEL = "_ZN3nfe25bwd_scatter_sorted_kernelILb1ELb1EEEvNS_4BwdKE:"
SLOGIC = re.compile(r"^\s*s_(and|or|andn2|orn2|xor|xnor)_b64\s+(vcc|s\[\d+:\d+\]),\s*([^,]+),\s*(\S+)")
CND = re.compile(r"^\s*v_cndmask_b32(_e32|_e64)?\s")
PAD = "\ts_nop 7\n\ts_nop 7\n"


def run(edit):
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    out, n = [], 0
    for i, l in enumerate(lines):
        if start < i < end:
            new, k = edit(l, lines, i)
            out += new
            n += k
        else:
            out.append(l)
    open(dst, "w").write("\n".join(out))
    print(f"patched {n} sites in {KERNEL[:-1]}")


def is_mask_logic(l):
    m = SLOGIC.match(l)
    return bool(m) and "exec" not in l


def reads_sgpr_mask(l):
    if not CND.match(l):
        return False
    ops = l.split(None, 1)[1]
    last = ops.split(",")[-1].strip()
    return last == "vcc" or last.startswith("s[") or "_e32" in l.split()[0]
