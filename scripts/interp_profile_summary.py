#!/usr/bin/env python3
"""Summary of scripts/interp_profile.sh: kernel-trace rows + HBM traffic per template point of warp_residual_interp_kernel from the PMC passes
(request-size counters: 128 / 64 / 32-byte reads, 64 / 32-byte writes — the accounting of profiles/summarize.py; FETCH_SIZE / WRITE_SIZE with the
guide's gfx950 correction beside it).  usage: interp_profile_summary.py <dir>"""
import glob
import json
import os
import sqlite3
import sys

src = sys.argv[1]


def db_of(sub):
    fs = sorted(glob.glob(os.path.join(src, sub, "*", "*_results.db")), key=os.path.getmtime)
    return sqlite3.connect(fs[-1]) if fs else None


bench = None
try:
    bench = json.loads([l for l in open(os.path.join(src, "trace_bench.json")) if l.startswith("{")][-1])
except Exception as e:  # noqa: BLE001
    print("(no bench line: %s)" % e)
print("BPVO_HIP_OPTIONS=lanes=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --pairs 256 --interp cubic --steps 2 --warmup 1 --cpu-pairs 0 --other-configs 0   (durations in us)")
db = db_of("trace")
if db:
    print("%-104s %8s %14s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:14]:
        print("%-104s %8d %14.1f %10.2f %7.2f" % (r[0][:104], r[1], r[2], r[3], r[4]))
if bench:
    print()
    print("bench.py line of the same run: value=%.0f GN it/s, ms_per_step=%.2f, roofline=%s" % (bench["value"], bench["ms_per_step"], json.dumps(bench["roofline"])))

per = {}
for sub in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(sub):
        continue
    d = db_of(os.path.basename(sub))
    if not d:
        continue
    for k, c, n, v, dur in d.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
        if "bpvo_hip" in k:
            per.setdefault(k, {})[c] = dict(launches=n, avg=v, avg_duration_ns=dur)
print()
for k, cs in per.items():
    if "warp_residual_interp" not in k and "irls_reduce" not in k:
        continue
    print(k[:120])
    for c, e in sorted(cs.items()):
        print("    %-28s %14.6g   (n=%d, avg kernel %.1f us)" % (c, e["avg"], e["launches"], (e["avg_duration_ns"] or 0) / 1e3))
    try:
        rd = 128 * cs["TCC_EA0_RDREQ_128B_sum"]["avg"] + 64 * cs["TCC_EA0_RDREQ_64B_sum"]["avg"] + 32 * cs["TCC_EA0_RDREQ_32B_sum"]["avg"]
        wr64 = cs["TCC_EA0_WRREQ_64B_sum"]["avg"]
        wr = 64 * wr64 + 32 * (cs["TCC_EA0_WRREQ_sum"]["avg"] - wr64)
        dur_s = cs["TCC_EA0_RDREQ_128B_sum"]["avg_duration_ns"] * 1e-9
        line = "    HBM bytes per launch by the request-size counters: read %.4g + write %.4g = %.4g -> %.0f GB/s over the pass's own average launch" % (rd, wr, rd + wr, (rd + wr) / dur_s / 1e9)
        if bench and "warp_residual_interp" in k:
            ppl = bench["roofline"]["points_per_launch"]
            line += "; %.1f B per template point (algorithmic %d B: every tap counted once per point)" % ((rd + wr) / ppl, bench["roofline"]["bytes_per_point"])
            line += "; %.3f of the 8 TB/s peak in counter bytes" % ((rd + wr) / dur_s / 8e12)
        print(line)
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            print("    guide-corrected FETCH_SIZE / WRITE_SIZE (KiB, x 2 for the 64-byte fetch granularity on gfx950): %.4g B per launch" % (1024.0 * (2.0 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"])))
    except KeyError as e:
        print("    (counter missing: %s)" % e)
