# kernels of ONE sequential addFrame (the last non-keyframe call of scripts/addframe_timeline.py): start offset, duration, gap to the next
W=${1:-perf_bitplanes}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
python3 $R/scripts/addframe_timeline.py $W
rm -rf /tmp/tra; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tra -- python3 $R/scripts/addframe_timeline.py $W > /tmp/tra.out 2>/tmp/tra.err
cat /tmp/tra.out
python3 - <<PY
import glob, sqlite3, os
fs = sorted(glob.glob("/tmp/tra/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
rows = [(n.split("(")[0].split("::")[-1], s, e) for n, s, e in db.execute("select name, start, end from kernels")]
try:
    rows += [("copy:" + str(n), s, e) for n, s, e in db.execute("select name, start, end from memory_copies")]
except Exception as ex:
    print("no memory_copies view:", ex)
rows.sort(key=lambda r: r[1])
# calls are separated by >= 1.5 ms of idle
calls = [[rows[0]]]
for r in rows[1:]:
    if r[1] - calls[-1][-1][2] > 1.5e6: calls.append([])
    calls[-1].append(r)
print(len(calls), "bursts; kernels per burst:", [len(c) for c in calls])
c = calls[-2]
t0 = c[0][1]
busy = sum(e - s for _, s, e in c)
print("burst: %d ops, span %.1f us, busy %.1f us" % (len(c), (c[-1][2] - t0) / 1e3, busy / 1e3))
for i, (n, s, e) in enumerate(c):
    gap = (c[i + 1][1] - e) / 1e3 if i + 1 < len(c) else 0.0
    print("%8.1f  %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, n[:70]))
PY
