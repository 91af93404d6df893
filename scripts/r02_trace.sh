#!/bin/bash
# GPU box: rocprofv3 kernel trace of one bench configuration -> gpurun_out/$1/trace_$2.txt   usage: r02_trace.sh OUT NAME PAIRS STEPS [ENV=VAL ...]
out=gpurun_out/$1; name=$2; pairs=$3; steps=$4; shift 4
mkdir -p $out; R=$(pwd); cache=/tmp/bpvo_cache; O=/tmp/bpvo_trace_$name; rm -rf $O
for kv in "$@"; do export "$kv"; done
timeout 300 python3 bench.py --pairs-per-gpu $pairs --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache $cache > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -- python3 $R/bench.py --pairs-per-gpu $pairs --steps $steps --warmup 1 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache $cache > $O.json 2> $O.err
cd $R
python3 - $O $O.json > $out/trace_$name.txt <<'PY'
import glob, json, os, sqlite3, sys
src = sys.argv[1]
fs = sorted(glob.glob(os.path.join(src, "*", "*_results.db")), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
tot = sum(r[2] for r in rows)
try:
    b = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print("bench: value %.0f, ms_per_step %.3f, steps %d -> wall %.1f ms; kernel time total %.1f ms" % (b["value"], b["ms_per_step"], b["steps"], b["ms_per_step"] * b["steps"], tot / 1e6))
except Exception as e:
    print("no bench json", e)
print("%-90s %8s %12s %10s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
for r in rows[:24]:
    print("%-90s %8d %12.1f %10.2f %6.2f" % (r[0][:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
PY
cat $out/trace_$name.txt | head -30
