# GN iterations/s and warp_residual roofline vs. pairs per GPU (bench.py defaults otherwise)
for n in ${@:-128 256 512}; do
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --pairs-per-gpu $n 2>&1 | tail -1 > /tmp/bs.json; python - <<PY
import json
d=json.load(open("/tmp/bs.json"))
r=d["roofline"]
print("pairs", $n, "value", round(d["value"]), "ms/step", round(d["ms_per_step"],2), "K6 GB/s", round(r["achieved"]), "frac", round(r["frac"],3), "pts/launch", round(r["points_per_launch"]), "synth_s", round(d["setup"]["synth_seconds"],1))
PY
done
