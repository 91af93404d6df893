# timeline of one HOST-buffer step of a 1024-pair batch: memory copies and kernels per stream in 10 ms buckets (rocprofv3 kernel + memory-copy trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 $R/scripts/shard_ab.py --host --pairs ${1:-1024} --steps 1 --warmup 1 --repeat 1 -- "" > /dev/null 2>&1     # renders the inputs
rm -rf /tmp/htl; timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/htl -- python3 $R/scripts/shard_ab.py --child /tmp/shard_ab_376x1241_${1:-1024}.npz --host --pairs ${1:-1024} --steps 1 --warmup 2 > /tmp/htl.json 2>/tmp/htl.err
tail -2 /tmp/htl.err; cat /tmp/htl.json
python3 - <<PY
import glob, sqlite3, os, collections
fs = sorted(glob.glob("/tmp/htl/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'cop' in t.lower() or 'kernel' in t.lower()][:12])
kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in kc else "queue_id"
K = list(db.execute("select name, start, end, %s from kernels order by start" % qcol))
mt = [t for t in tabs if t.lower() in ("memory_copies", "memory_copy")]
mc = [r[1] for r in db.execute("pragma table_info(%s)" % mt[0])] if mt else []
print("copy columns:", mc)
M = list(db.execute("select start, end, size from %s order by start" % mt[0])) if mt else []
# the last step = after the last big gap in copies: take the last 1/2 of the run by locating the last ingest burst
ing = [r for r in K if "ingest" in r[0]]
# warm-up steps + the timed one: the timed step starts at the first big H2D copy after the last pause of more than 20 ms between such copies
big = [m for m in M if m[2] > 1000000]
t0 = big[0][0]
for a, b in zip(big, big[1:]):
    if b[0] - a[1] > 20e6: t0 = b[0]
tend = max(r[2] for r in K)
print("timed step: %.1f ms" % ((tend - t0) / 1e6))
B = 10e6
nb = int((tend - t0) / B) + 1
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0].replace("_kernel", "")
qs = sorted(set(r[3] for r in K if r[1] >= t0))
print("bucket(ms)  H2D MB   " + "  ".join("q%s busy%%/top" % q for q in qs))
for b in range(nb):
    lo, hi = t0 + b * B, t0 + (b + 1) * B
    mb = sum(m[2] * max(0, min(m[1], hi) - max(m[0], lo)) / max(1, m[1] - m[0]) for m in M if m[1] > lo and m[0] < hi) / 1e6
    cells = []
    for q in qs:
        acc = collections.Counter()
        for n, s, e, qq in K:
            if qq == q and e > lo and s < hi: acc[short(n)] += min(e, hi) - max(s, lo)
        tot = sum(acc.values())
        top = acc.most_common(2)
        cells.append("%3d%% %s" % (100 * tot / B, ",".join("%s:%d" % (k[:14], 100 * v / B) for k, v in top)))
    print("%4d-%4d  %7.0f   %s" % (b * 10, b * 10 + 10, mb, "  |  ".join(cells)))
PY
