# where the time of one Gauss-Newton iteration goes at batch size 1: kernel durations and the gaps between consecutive
# kernels (end -> next start) from a rocprofv3 kernel trace.  bash scripts/b1_timeline.sh [pairs]
P=${1:-1}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trb; timeout 400 rocprofv3 --kernel-trace -d /tmp/trb -- python3 $R/bench.py --pairs-per-gpu $P --steps 4 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trb.json 2>/tmp/trb.err
python3 - <<PY
import glob, sqlite3, os, json, collections
fs = sorted(glob.glob("/tmp/trb/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
rows = list(db.execute("select name, start, end from kernels order by start"))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (n, s, e) in enumerate(rows):
    dur[short(n)].append((e - s) / 1e3)
    if i + 1 < len(rows):
        gap[short(n) + " -> " + short(rows[i + 1][0])].append((rows[i + 1][1] - e) / 1e3)
print("kernel durations (us): name, launches, mean, median")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    v2 = sorted(v); print("  %-28s %6d %8.2f %8.2f" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2]))
print("gaps end -> next start (us): pair, count, mean, median")
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:10]:
    v2 = sorted(v); print("  %-50s %6d %8.2f %8.2f" % (k, len(v), sum(v) / len(v), v2[len(v2) // 2]))
d = json.loads(open("/tmp/trb.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "GN it/s; ms/step", round(d["ms_per_step"], 3), "; iterations/step", d["gn_iterations_per_step"], "-> us per iteration", round(1e3 * d["ms_per_step"] / d["gn_iterations_per_step"], 2))
PY
