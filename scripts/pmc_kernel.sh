# PMC counters of the kernels whose name contains $1, one rocprofv3 pass per counter set ($2, $3, ...: quoted, space-separated names),
# on a short fixed-iteration 128-pair bench.   bash scripts/pmc_kernel.sh bitplanes "SQ_WAVE_CYCLES SQ_BUSY_CYCLES ..." "SQ_INSTS_LDS ..."
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
R=/root/repo
CACHE=/tmp/bpvo_bench_inputs_128
ARGS="--pairs-per-gpu 128 --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --fixed-iters 2"
BPVO_HIP_OPTIONS=lanes=1 timeout 300 python3 $R/bench.py $ARGS --input-cache $CACHE > /dev/null 2>&1
i=0
for CNT in "$@"; do
  i=$((i+1)); rm -rf /tmp/pk$i
  BPVO_HIP_OPTIONS=lanes=1 timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/pk$i -- python3 $R/bench.py $ARGS --gen-workers 1 --input-cache $CACHE > /tmp/pk$i.json 2> /tmp/pk$i.err
  python3 - <<PY
import glob, sqlite3, os
fs = sorted(glob.glob("/tmp/pk$i/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%$PAT%' group by kernel_name, counter_name"
for k, c, n, v, dur in db.execute(q):
    print("%-50s %-28s %16.6g  (n=%d, avg %.1f us)" % (k.split("(")[0][-50:], c, v, n, (dur or 0) / 1e3))
PY
done
