#!/bin/bash
# rocprofv3 evidence for the non-kLinear warp + residual kernel (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash scripts/interp_profile.sh r06'
# 256 pairs of the headline shape (1241x376 bit-planes, Tukey, 4 levels) with kCubic, one lane: kernel trace + stats, then the HBM
# request counters in passes of their own (--pmc with --kernel-trace only).  Writes gpurun_out/<tag>_interp_kernel.txt.
set -u
TAG=${1:-r06}
R=$(pwd)
O=/tmp/bpvo_interp_$TAG
rm -rf "$O"; mkdir -p "$O" "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
CACHE=/tmp/bpvo_bench_inputs_interp
ARGS="--pairs 256 --interp cubic --steps 2 --warmup 1 --cpu-pairs 0 --other-configs 0"
export BPVO_HIP_OPTIONS=lanes=1
timeout 300 python3 "$R/bench.py" $ARGS --no-profile --input-cache $CACHE > /dev/null 2> "$O/cache.err"; echo "inputs rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d "$O/trace" -- python3 "$R/bench.py" $ARGS --gen-workers 1 --input-cache $CACHE > "$O/trace_bench.json" 2> "$O/trace.err"; echo "trace rc=$?"
i=0
for CNT in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d "$O/pmc$i" -- python3 "$R/bench.py" $ARGS --no-profile --gen-workers 1 --input-cache $CACHE > "$O/pmc$i.json" 2> "$O/pmc$i.err"
  echo "pmc$i ($CNT) rc=$?"
done
cd "$R" && python3 scripts/interp_profile_summary.py "$O" > "$R/gpurun_out/${TAG}_interp_kernel.txt"; tail -30 "$R/gpurun_out/${TAG}_interp_kernel.txt"
