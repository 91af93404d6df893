# rocprofv3 kernel trace of a short bench run; prints the top kernels (avg us per launch)
P=${1:-256}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trk; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/trk -- python3 $R/bench.py --pairs-per-gpu $P --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trk.json 2>/tmp/trk.err
python3 - <<PY
import glob, sqlite3, os
fs = sorted(glob.glob("/tmp/trk/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-90s %6d %12.1f %10.2f %6.2f" % (r[0][:90], r[1], r[2]/1e3 if r[2] > 1e7 else r[2], r[3]/1e3 if r[3] > 1e6 else r[3], r[4]))
PY
