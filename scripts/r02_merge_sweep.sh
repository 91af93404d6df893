#!/bin/bash
# GPU box: irls_reduce merged vs split launches, lanes, over batch sizes
out=gpurun_out/${1:-r02f}; mkdir -p $out
cache=/tmp/bpvo_cache
run() { local name=$1 pairs=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --pairs-per-gpu $pairs --steps $steps --warmup 1 --cpu-pairs 0 --other-configs 0 --input-cache $cache > $out/bench_$name.json 2> $out/bench_$name.err
  python - $out/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("%-28s %9.0f GN it/s  %8.2f ms/step  K6 frac %.3f (%.1f us)  gn_loop frac %.3f  tap %.4f/%.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("frac", 0), 1e3 * r.get("avg_launch_ms", 0), (d.get("gn_loop_roofline") or {}).get("frac", 0), d["tap_cache"]["hit_rate"] or 0, d["tap_cache"]["first_8_linearisations_of_a_level"]["hit_rate"] or 0))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for pairs in 1024 512 256 128 32 8 1; do
  steps=$([ $pairs -ge 512 ] && echo 3 || echo 10)
  run split_$pairs $pairs $steps BPVO_HIP_IRLS_MERGE_BELOW=0
  run merged_$pairs $pairs $steps BPVO_HIP_IRLS_MERGE_BELOW=100000
done
for pairs in 1024 128 32; do
  steps=$([ $pairs -ge 512 ] && echo 3 || echo 10)
  run merged_lanes2_$pairs $pairs $steps BPVO_HIP_IRLS_MERGE_BELOW=100000 BPVO_HIP_LANES=2
  run split_lanes2_$pairs $pairs $steps BPVO_HIP_IRLS_MERGE_BELOW=0 BPVO_HIP_LANES=2
  run merged_lanes4_$pairs $pairs $steps BPVO_HIP_IRLS_MERGE_BELOW=100000 BPVO_HIP_LANES=4
done
