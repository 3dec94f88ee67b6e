#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing: isa_stats.py file.s <substring>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % sys.argv[2], l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ins = []
for l in lines[start + 1:end + 1]:
    t = l.strip()
    if not t or t[0] in ".;/" or t.endswith(":"):
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
print("total", len(ins))
cls = collections.Counter()
for k, v in c.items():
    if k.startswith(("global_", "buffer_", "flat_", "scratch_")):
        cls["vmem:" + k] += v
    elif k.startswith("ds_"):
        cls["lds:" + k] += v
    elif k.startswith("v_"):
        cls["valu"] += v
    elif k.startswith("s_waitcnt"):
        cls["s_waitcnt"] += v
    elif k.startswith("s_cbranch") or k.startswith("s_branch"):
        cls["branch"] += v
    elif k.startswith("s_load"):
        cls["smem"] += v
    elif k.startswith("s_"):
        cls["salu"] += v
    else:
        cls[k] += v
for k, v in sorted(cls.items(), key=lambda x: -x[1]):
    print(f"{v:6d} {k}")
print("top valu:", [(k, v) for k, v in c.most_common(60) if k.startswith("v_")][:18])
