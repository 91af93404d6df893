#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash profiles/collect_profiles.sh r01'
# 1. kernel trace + stats of the default bench command (the default estimation lanes, the run `value` comes from) and of the same
#    command with BPVO_HIP_OPTIONS=lanes=1 (the per-launch durations bench.py's single-lane roofline pass is compared with);
# 2. PMC passes (counters in their own runs, --kernel-trace only) on the BENCHED workload itself — 1024 pairs, converge mode,
#    2 steps, one lane — and on the timing-tolerance batch (conf/perf_*.cfg tolerances, 3 levels); synthetic pairs read from a
#    cache rendered beforehand, each pass under its own timeout.
# Raw output goes to gpurun_out/profiles_<tag>/; profiles/summarize.py turns it into the committed summaries.
set -u
TAG=${1:-r02}
R=$(pwd)
# raw rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back); only the summaries are copied out
O=/tmp/bpvo_profiles_$TAG
rm -rf "$O"; mkdir -p "$O"; mkdir -p "$R/gpurun_out/profiles_$TAG"
cd /tmp && export TMPDIR=/tmp
# the synthetic inputs are rendered once, outside the profiler (a fork pool under rocprofv3 hangs: the profiler initialises
# the GPU before python starts), and the profiled runs read them back from /tmp
CACHE=/tmp/bpvo_bench_inputs
timeout 300 python3 "$R/bench.py" --steps 1 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --input-cache $CACHE > /dev/null 2> "$O/cache_default.err"; echo "inputs rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/trace" -- python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --gen-workers 1 --input-cache $CACHE \
    > "$O/trace_bench.json" 2> "$O/trace.err"; echo "trace rc=$?"
export BPVO_HIP_OPTIONS=lanes=1
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/trace1" -- python3 "$R/bench.py" --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --gen-workers 1 --input-cache $CACHE \
    > "$O/trace1_bench.json" 2> "$O/trace1.err"; echo "trace (one lane) rc=$?"
i=0
for CNT in "FETCH_SIZE" "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES" \
           "TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d "$O/pmc$i" -- python3 "$R/bench.py" \
      --steps 2 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache $CACHE > "$O/pmc$i.json" 2> "$O/pmc$i.err"
  echo "pmc$i ($CNT) rc=$?"
done
# the same request-size counters on the timing-tolerance batch (3 levels, ~7x fewer iterations per level: the tap cache is colder)
j=0
for CNT in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
  j=$((j+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d "$O/tpmc$j" -- python3 "$R/bench.py" --tolerances timing --levels 3 \
      --steps 2 --warmup 0 --cpu-pairs 0 --other-configs 0 --no-profile --gen-workers 1 --input-cache $CACHE > "$O/tpmc$j.json" 2> "$O/tpmc$j.err"
  echo "tpmc$j ($CNT) rc=$?"
done
cd "$R" && python3 profiles/summarize.py "$O" "$TAG" && cp "$O"/${TAG}_* "$O"/trace_bench.json "$O"/trace1_bench.json "$O"/pmc*.json "$O"/tpmc*.json "$O"/*.err "$R/gpurun_out/profiles_$TAG/"
