#!/bin/bash
# persistent GN kernel vs the four-kernel chain for small groups: GN it/s by pairs per call; sequential addFrame
out=gpurun_out/r02p; mkdir -p $out
for n in 1 2 4 8; do
  for on in 0 1; do
    BPVO_HIP_PERSISTENT=$on BPVO_HIP_PERSIST_MAX_WS=8 timeout 300 python bench.py --steps 8 --warmup 2 --cpu-pairs 0 --other-configs 0 --no-profile --pairs-per-gpu $n 2>/dev/null | tail -1 > /tmp/ps.json
    python - <<PY
import json
d=json.load(open("/tmp/ps.json"))
print("pairs $n persistent $on: %.0f GN it/s, %.3f ms per step" % (d["value"], d["ms_per_step"]))
PY
  done
done 2>&1 | tee $out/sweep.txt
for on in 0 1; do
  echo "== addFrame phases, BPVO_HIP_PERSISTENT=$on"
  BPVO_HIP_PERSISTENT=$on timeout 300 python scripts/addframe_phases.py 2>&1 | tail -4
done 2>&1 | tee $out/addframe.txt
