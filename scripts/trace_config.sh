# rocprofv3 kernel trace of an arbitrary bench configuration: bash scripts/trace_config.sh <bench args...>
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/trc; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/trc -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 "$@" > /tmp/trc.json 2>/tmp/trc.err
python3 - <<PY
import glob, sqlite3, os, json
fs = sorted(glob.glob("/tmp/trc/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
for r in list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))[:14]:
    print("%-100s %6d %12.1f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))
d = json.loads(open("/tmp/trc.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "points/step", d["points_linearized_rank0"] / d["steps"])
PY
