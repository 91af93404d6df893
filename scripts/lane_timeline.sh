# timeline of one step of a batch: per HIP stream (lane) the start of the frame stage, of every pyramid level (level_begin launches) and
# the end of the last kernel; plus the fraction of the step during which at least one kernel was running.  bash scripts/lane_timeline.sh [pairs]
P=${1:-128}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trl; timeout 400 rocprofv3 --kernel-trace -d /tmp/trl -- python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trl.json 2>/tmp/trl.err
tail -2 /tmp/trl.err
python3 - <<PY
import glob, sqlite3, os, collections
fs = sorted(glob.glob("/tmp/trl/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
rows = list(db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0]
# the last step: starts at the last ingest kernel that follows a gap
ing = [i for i, r in enumerate(rows) if "ingest" in r[0]]
nl = len(set(r[3] for r in rows if "ingest" in r[0]))
first = ing[-nl] if len(ing) >= nl else ing[0]
rows = rows[first:]
t0 = rows[0][1]
print("step: %d kernels, %.2f ms" % (len(rows), (max(r[2] for r in rows) - t0) / 1e6))
byq = collections.defaultdict(list)
for r in rows: byq[r[3]].append(r)
for q, rs in byq.items():
    marks = []
    for n, s, e, _ in rs:
        sn = short(n)
        if sn in ("ingest_kernel", "level_begin_kernel", "saliency_kernel", "template_build_kernel", "pack_records_kernel"):
            if not marks or marks[-1][0] != sn or sn == "level_begin_kernel": marks.append((sn, (s - t0) / 1e6))
    busy = sum(e - s for _, s, e, _ in rs) / 1e6
    print("queue", q, "kernels", len(rs), "busy %.2f ms" % busy, " ".join("%s@%.2f" % (m[0].replace("_kernel", ""), m[1]) for m in marks), "end@%.2f" % ((rs[-1][2] - t0) / 1e6))
# union of busy intervals
iv = sorted((s, e) for _, s, e, _ in rows)
tot = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: tot += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
tot += ce - cs
print("some kernel running: %.2f ms of %.2f" % (tot / 1e6, (max(r[2] for r in rows) - t0) / 1e6))
# frame stage (everything before the first level_begin of a queue): time per kernel
for q, rs in byq.items():
    acc = collections.OrderedDict()
    for n, s, e, _ in rs:
        sn = short(n).replace("_kernel", "")
        if sn == "level_begin": break
        a = acc.setdefault(sn, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
    if acc: print("queue", q, "frame stage:", " ".join("%s %dx=%.0fus" % (k, v[0], v[1]) for k, v in acc.items()), "| total %.2f ms" % (sum(v[1] for v in acc.values()) / 1e3))
# per queue and level: launches, mean duration per kernel, summed gaps (end -> next start on the same queue)
for q, rs in byq.items():
    lvl = -1; acc = collections.OrderedDict()
    for i, (n, s, e, _) in enumerate(rs):
        sn = short(n).replace("_kernel", "")
        if sn == "gn_pipe":      # gn_pipe_kernel<LOSS, WIDE, NARROW>: W/I = warp_residual / irls_reduce, M/G = median / gn_step
            a3 = n.split("<")[1].split(">")[0].replace(" ", "").split(",")
            sn = "pipe_" + "WI"[int(a3[1])] + "MG"[int(a3[2])]
        if sn == "level_begin": lvl += 1
        if lvl < 0: continue
        a = acc.setdefault(lvl, {"k": collections.defaultdict(list), "gap": 0.0, "t0": s, "t1": e})
        a["k"][sn].append((e - s) / 1e3); a["t1"] = e
        if i + 1 < len(rs): a["gap"] += max(0, rs[i + 1][1] - e) / 1e3
    for l, a in acc.items():
        ks = " ".join("%s %dx%.1f" % (k, len(v), sum(v) / len(v)) for k, v in a["k"].items() if k in ("warp_residual", "median_finish", "irls_reduce_both", "irls_reduce", "gn_step") or k.startswith("pipe_"))
        print("queue", q, "level#", l, "span %.2f ms" % ((a["t1"] - a["t0"]) / 1e6), "kernel time %.2f ms" % (sum(sum(v) for v in a["k"].values()) / 1e3), "gaps %.2f ms |" % (a["gap"] / 1e3), ks)
# per-ms: number of distinct kernels by class
PY
