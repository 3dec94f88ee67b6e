#!/usr/bin/env python3
"""Per-kernel resource table of the product translation unit (VGPRs, SGPRs, scratch, LDS, the compiler's waves-per-SIMD)
from `hipcc -Rpass-analysis=kernel-resource-usage` -- `make -C fots.pytorch_amd/csrc resources` piped through this.
    python tools/kernel_resources.py [extra hipcc flags]  ->  one line per kernel, sorted by name"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
       "-I" + os.path.join(ROOT, "include"), "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
       os.path.join(ROOT, "fots.pytorch_amd", "csrc", "rroi_align_hip.hip")] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = {}, None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
for mangled, name in sorted(zip(rows, names), key=lambda t: t[1]):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name).split("(")[0]
    r = rows[mangled]
    print("%-86s vgpr %3d sgpr %3d scratch %3d lds %6d occ %d" % (name, r.get("vgpr", -1), r.get("sgpr", -1), r.get("scratch", -1),
                                                              r.get("lds", -1), r.get("occ", -1)))
