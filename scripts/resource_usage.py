#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per function (kernels and noinline device functions).

  hipcc --offload-arch=gfx950 -O3 ... -c kernels_gn.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2> res.txt
  python scripts/resource_usage.py res.txt [filter]
"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for line in txt.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+: (?:Function Name|Name): (\S+)", line) or re.search(r"(?:Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Dynamic Stack", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\S+)", line)
        if m and cur is not None and key not in cur:
            cur[key] = m.group(1)
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("bpvo_hip::", "").replace("void ", "")
    if flt and flt not in d:
        continue
    print("%-70s vgpr %4s sgpr %4s scratch %5s occ %2s lds %7s" % (d[:70], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
                                                                  r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
