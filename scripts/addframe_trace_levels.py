#!/usr/bin/env python3
"""Per-dispatch view of scripts/addframe_trace.sh's kernel trace: for every Gauss-Newton kernel the launches grouped by grid size (= pyramid
level of the single pair) with count / average / max duration, and the first N dispatches in order.  usage: addframe_trace_levels.py <dir> [N [skip]]"""
import glob
import sqlite3
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*_results.db"))[-1]
n_first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
db = sqlite3.connect(f)
try:
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else "grid_size")
    rows = list(db.execute(f"select name, start, end, {gx} from kernels order by start"))
except Exception as e:  # noqa: BLE001
    print("schema:", [r for r in db.execute("select name, type from sqlite_master")], e)
    sys.exit(1)
agg = {}
last_warp = 0
for i, (name, s, e, g) in enumerate(rows):
    if "warp_residual" in name:
        last_warp = g
    elif "median_finish" in name:          # one grid for every level: label it with the grid of the warp + residual launch before it
        rows[i] = (name, s, e, last_warp)
        g = last_warp
    k = (name.split("(")[0].replace("void bpvo_hip::", "").replace("bpvo_hip::", "")[:60], g)
    a = agg.setdefault(k, [0, 0.0, 0.0, []])
    a[3].append((e - s) / 1e3)
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] = max(a[2], (e - s) / 1e3)
print("%-62s %10s %7s %10s %9s %9s  %s" % ("kernel", "grid_x", "calls", "total_us", "avg_us", "max_us", "deciles of the duration (us)"))
for (k, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    d = sorted(a[3])
    print("%-62s %10d %7d %10.1f %9.2f %9.2f  %s" % (k, g, a[0], a[1], a[1] / a[0], a[2], " ".join("%.0f" % d[min(len(d) - 1, len(d) * q // 10)] for q in range(1, 10))))
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = rows[skip][1] if len(rows) > skip else 0
prev_end = t0
for name, s, e, g in rows[skip:skip + n_first]:
    print("gap %6.2f " % ((s - prev_end) / 1e3), end="")
    prev_end = e
    print("%10.1f us  +%8.2f  grid %8d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, g, name.split("(")[0][-70:]))
