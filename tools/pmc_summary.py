#!/usr/bin/env python3
"""Fold the per-pass rocprofv3 CSVs (counter_collection + kernel_trace) into one table:
rows = kernels, columns = counters (mean per dispatch) + mean duration."""
import glob
import os
import sys

import pandas as pd

root = sys.argv[1]
cnt = {}
dur = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    df = pd.read_csv(f)
    name_col = "Kernel_Name" if "Kernel_Name" in df.columns else "Kernel Name"
    for (k, c), g in df.groupby([name_col, "Counter_Name"]):
        per_dispatch = g.groupby("Dispatch_Id")["Counter_Value"].sum()
        cnt.setdefault(k, {})[c] = per_dispatch.mean()
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    df = pd.read_csv(f)
    df["dur"] = df["End_Timestamp"] - df["Start_Timestamp"]
    for k, g in df.groupby("Kernel_Name"):
        dur.setdefault(k, []).append(g["dur"].mean() / 1e3)
rows = []
for k, d in cnt.items():
    short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f"## {short}   avg duration (profiled) {sum(dur.get(k, [0])) / max(1, len(dur.get(k, [0]))):.1f} us")
    for c in sorted(d):
        print(f"  {c:44s} {d[c]:16.1f}")
    print()
