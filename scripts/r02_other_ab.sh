#!/bin/bash
out=gpurun_out/${1:-r02h}; mkdir -p $out
run() { local name=$1; shift
  env "$@" timeout 900 python bench.py --pairs-per-gpu 128 --steps 10 --warmup 2 --cpu-pairs 0 --other-configs 128 --input-cache /tmp/bpvo_cache > $out/ab_$name.json 2> $out/ab_$name.err
  python - $out/ab_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    o = d["other_configs"]
    print("%-22s main128 %7.0f |" % (sys.argv[2], d["value"]), " | ".join("%s %.0f (%.2f ms)" % (k[:28], v.get("value", 0), v.get("ms_per_step", v.get("ms_per_frame", 0))) for k, v in o.items()))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run default
run sync_rounds BPVO_HIP_SYNC_ROUNDS=1
run noprof BPVO_BENCH_OTHER_PROFILING=0
run lanes1 BPVO_HIP_LANES=1
run sync_noprof BPVO_HIP_SYNC_ROUNDS=1 BPVO_BENCH_OTHER_PROFILING=0
