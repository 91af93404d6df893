#!/bin/bash
# GPU box: launch-chain sweep (four launches vs fused tails) over batch sizes; one JSON line per run under gpurun_out/$1/
out=gpurun_out/${1:-r02b}; mkdir -p $out
cache=/tmp/bpvo_cache
run() { # name pairs steps env...
  local name=$1 pairs=$2 steps=$3; shift 3
  env "$@" timeout 600 python bench.py --pairs-per-gpu $pairs --steps $steps --warmup 1 --cpu-pairs 0 --other-configs 0 --input-cache $cache > $out/bench_$name.json 2> $out/bench_$name.err
  python - $out/bench_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("%-28s %9.0f GN it/s  %8.2f ms/step  K6 frac %.3f (%.1f us, %.0f pts)  med %s tap %s" % (sys.argv[2], d["value"], d["ms_per_step"], r.get("frac", 0), 1e3 * r.get("avg_launch_ms", 0), r.get("points_per_launch", 0), d.get("median_selections"), d.get("tap_cache")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for pairs in 1024 128; do
  steps=$([ $pairs = 1024 ] && echo 3 || echo 10)
  run classic_$pairs $pairs $steps BPVO_HIP_CHAIN=classic
  run tails_$pairs $pairs $steps BPVO_HIP_CHAIN=tails
done
run classic_lanes2_128 128 10 BPVO_HIP_CHAIN=classic BPVO_HIP_LANES=2
run tails_lanes2_128 128 10 BPVO_HIP_CHAIN=tails BPVO_HIP_LANES=2
run tails_lanes2_1024 1024 3 BPVO_HIP_CHAIN=tails BPVO_HIP_LANES=2
for pairs in 32 8 1; do
  run classic_$pairs $pairs 10 BPVO_HIP_CHAIN=classic
  run tails_$pairs $pairs 10 BPVO_HIP_CHAIN=tails
done
