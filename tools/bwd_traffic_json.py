#!/usr/bin/env python3
"""profiles/bwd_traffic.json (what bench.py's extra.backward.roofline reports) from the summary of the backward's
counter passes (tools/profile_bwd.sh -> pmc_bwd_traffic.md, written by tools/pmc_summary.py).
FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies a 128-byte read request as 64 B, so FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM traffic section); WRITE_SIZE as is.     python tools/bwd_traffic_json.py <md> <round>"""
import json
import re
import sys

md, rnd = sys.argv[1], sys.argv[2]
kern, cur = {}, None
for line in open(md):
    m = re.match(r"## (\S.*?)\s+avg duration \(profiled\) ([0-9.]+) us", line)
    if m:
        cur = m.group(1) if m.group(1).startswith("rroi_") else None
        if cur:
            kern[cur] = {"us": float(m.group(2))}
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)", line)
    if m and cur:
        kern[cur][m.group(1)] = float(m.group(2))
out = {
    "source": f"round {rnd}: separate rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bwd_profile.py "
              f"(profiles/r{int(rnd):02d}_pmc_bwd_traffic.md; kernel-trace stats of the unprofiled-counter run: "
              f"profiles/r{int(rnd):02d}_bwd_kernel_stats.csv), tools/profile_bwd.sh, MI355X; FETCH_SIZE doubled (gfx950 "
              "tallies 128-byte read requests at 64 B), WRITE_SIZE as is",
    "what": "rroi_align_backward_hip, PATH_TILED (one-pass bucket lists, NCHW gradient written by the gather), BASELINE configs[2]",
    "kernel_us": {k: v["us"] for k, v in kern.items()},
    "traffic_per_kernel": {k: {"read_bytes": int(v.get("FETCH_SIZE", 0) * 2048), "written_bytes": int(v.get("WRITE_SIZE", 0) * 1024)}
                           for k, v in kern.items()},
    "kernel_us_how": "average duration of each launch in the --pmc passes over the cfg3 call alone (RROI_BWD_ONLY=1; "
                     "profiled clocks run ~3 % below unprofiled ones)",
}
out["traffic_bytes_per_call"] = sum(v["read_bytes"] + v["written_bytes"] for v in out["traffic_per_kernel"].values())
print(json.dumps(out, indent=1))
