#!/usr/bin/env python3
"""Compact per-kernel resource table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage):
    python tools/resource_usage.py nerffaceediting_amd/csrc/nfe_render.hip [extra hipcc flags...]"""
import re
import subprocess
import sys

src, flags = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Iinclude", "-Inerffaceediting_amd/csrc",
       *flags, "-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'vspill':>6s} {'sspill':>6s} {'LDS':>7s} {'occ':>4s}")
for r, d in zip(rows, demangle):
    d = re.sub(r"\(.*\)$", "", d).replace("void ", "").replace("nfe::", "")
    print(f"{d[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>8s} "
          f"{r.get('VGPRs Spill', r.get('VGPR Spill','?')):>6s} {r.get('SGPRs Spill', r.get('SGPR Spill','?')):>6s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
