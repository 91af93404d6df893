# the frame stage of one step, launch by launch (rocprofv3 kernel trace, ONE estimation lane so that nothing else runs beside it):
# bash scripts/frame_stage_trace.sh [pairs]
P=${1:-1024}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
export BPVO_HIP_OPTIONS=lanes=1
timeout 300 python3 $R/bench.py --pairs $P --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache /tmp/bpvo_bench_inputs > /dev/null 2>&1
rm -rf /tmp/trf; timeout 400 rocprofv3 --kernel-trace -d /tmp/trf -- python3 $R/bench.py --pairs $P --steps 1 --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs > /tmp/trf.json 2>/tmp/trf.err
tail -2 /tmp/trf.err
python3 - <<PY
import glob, sqlite3, os
fs = sorted(glob.glob("/tmp/trf/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
rows = list(db.execute("select name, start, end from kernels order by start"))
short = lambda n: n.split("(")[0].split("::")[-1]
ing = [i for i, r in enumerate(rows) if "ingest" in r[0]]
rows = rows[ing[-1]:]
t0 = rows[0][1]
tot = 0.0
for n, s, e in rows:
    sn = short(n)
    if "level_begin" in sn: break
    print("%8.3f ms  %9.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, sn))
    tot += (e - s) / 1e3
print("frame stage: %.2f ms of kernels, ends at %.2f ms" % (tot / 1e3, (s - t0) / 1e6))
PY
