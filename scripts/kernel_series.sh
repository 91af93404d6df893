# per-launch durations (us) of one kernel in launch order over the last step of a traced bench run
P=${1:-128}; PAT=${2:-median_finish}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trs; timeout 400 rocprofv3 --kernel-trace -d /tmp/trs -- python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trs.json 2>/tmp/trs.err
python3 - <<PY
import glob, sqlite3, os
fs = sorted(glob.glob("/tmp/trs/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
v = [t for t in tabs if t == "kernels"] 
rows = list(db.execute("select name, start, end from kernels order by start"))
sel = [(e - s) / 1e3 for n, s, e in rows if "$PAT" in n]
half = sel[len(sel)//2:]
print(len(sel), "launches; second half (timed step):")
print(" ".join("%.0f" % x for x in half))
PY
