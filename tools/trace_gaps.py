#!/usr/bin/env python3
"""Timeline of consecutive kernels from a rocprofv3 --kernel-trace CSV: for the last N repetitions of a
repeating kernel sequence, the average duration of each kernel and the average idle gap before it."""
import csv
import sys
from collections import defaultdict

import re


def short(name):
    m = re.search(r"(rroi_\w+|[A-Za-z_]\w*)(?=[<(]|$)", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    return (m.group(1) if m else name)[:60]


path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "rroi_affine_kernel"
rows = []
with open(path) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
# split into calls at the anchor kernel
calls, cur = [], []
for s, e, n in rows:
    if anchor in n and cur:
        calls.append(cur)
        cur = []
    cur.append((s, e, n))
calls.append(cur)
shape = tuple(n for _, _, n in calls[-1])
same = [c for c in calls if tuple(n for _, _, n in c) == shape][-20:]
print(f"{len(same)} calls of {len(shape)} kernels")
dur = defaultdict(float)
gap = defaultdict(float)
span = 0.0
for ci, c in enumerate(same):
    for i, (s, e, n) in enumerate(c):
        dur[i] += (e - s) / 1e3
        if i:
            gap[i] += (s - c[i - 1][1]) / 1e3
    span += (c[-1][1] - c[0][0]) / 1e3
for i, n in enumerate(shape):
    print(f"{i} {n:60s} gap before {gap[i] / len(same):7.2f} us   duration {dur[i] / len(same):7.2f} us")
print(f"first start -> last end: {span / len(same):.2f} us; sum of durations {sum(dur.values()) / len(same):.2f} us")
if len(same) > 1:
    per = (same[-1][0][0] - same[0][0][0]) / 1e3 / (len(same) - 1)
    print(f"call period (start to start): {per:.2f} us")
