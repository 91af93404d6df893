# staggered lanes (batch_run_staggered) against the stage-by-stage batch, per batch size and lane count
for n in ${@:-128 1024}; do
for cfg in "0 2" "1 2" "1 3" "1 4"; do
set -- $cfg
BPVO_HIP_STAGGER=$1 BPVO_HIP_LANES=$2 timeout 600 python bench.py --steps 6 --warmup 2 --cpu-pairs 0 --other-configs 0 --pairs-per-gpu $n 2>&1 | tail -1 > /tmp/bs.json; python - <<PY
import json
d=json.load(open("/tmp/bs.json"))
print("pairs", $n, "stagger", $1, "lanes", $2, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2))
PY
done
done
