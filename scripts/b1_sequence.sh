# the kernel sequence of ONE step at batch size P (default 1): start (relative), duration, queue of every kernel of the last step in a
# rocprofv3 kernel trace — what runs next to what, and which gaps are on the critical path.  bash scripts/b1_sequence.sh [pairs] [options]
P=${1:-1}
export BPVO_HIP_OPTIONS=${2:-}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 300 python3 $R/bench.py --pairs-per-gpu $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trs; timeout 400 rocprofv3 --kernel-trace -d /tmp/trs -- python3 $R/bench.py --pairs-per-gpu $P --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trs.json 2>/tmp/trs.err
python3 - <<PY
import glob, sqlite3, os, json
fs = sorted(glob.glob("/tmp/trs/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, end, queue_id || '/s' || stream_id from kernels order by start"))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0]
# the last step: from the last ingest_kernel on
i0 = max(i for i, r in enumerate(rows) if "ingest" in r[0])
t0 = rows[i0][1]
prev_end = t0
for n, s, e, qid in rows[i0:]:
    print("%9.1f  +%7.1f us  gap %7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, qid, short(n)))
    prev_end = max(prev_end, e)
d = json.loads(open("/tmp/trs.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), "GN it/s; ms/step", round(d["ms_per_step"], 3), "; iterations/step", d["gn_iterations_per_step"])
PY
