#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table from hipcc's kernel-resource-usage remarks.
usage: tools/kernel_resources.py ringsnark_amd/csrc/witness.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+([A-Za-z /\[\]]+): (\d+)", line) or re.search(r":\s{4}([A-Za-z /\[\]]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt in dem:
        print("%-60s vgpr %3d agpr %3d sspill %3d vspill %3d occ %d" % (dem[:60], r.get("VGPRs", -1), r.get("AGPRs", -1),
              r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1)))
