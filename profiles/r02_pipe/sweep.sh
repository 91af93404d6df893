# pipelined chain (gn_pipe_kernel) against the four-kernel chain, per batch size, lane count and stagger
for n in ${@:-128 1024}; do
for cfg in "0 2 1" "1 1 0" "1 2 1" "1 2 0"; do
set -- $cfg
BPVO_HIP_PIPE=$1 BPVO_HIP_LANES=$2 BPVO_HIP_STAGGER=$3 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --other-configs 0 --pairs-per-gpu $n 2>&1 | tail -1 > /tmp/bs.json; python - <<PY
import json
d=json.load(open("/tmp/bs.json"))
print("pairs", $n, "pipe", $1, "lanes", $2, "stagger", $3, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2))
PY
done
done
