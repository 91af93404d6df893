for ppb in 512 1024 2048; do
BPVO_HIP_PPB=$ppb timeout 300 python bench.py --steps 5 --warmup 1 --cpu-pairs 0 --profile-all 2>&1 | tail -1 > /tmp/pa.json; python - <<PY
import json
d=json.load(open("/tmp/pa.json"))
k=d["kernels"]
print("ppb", $ppb, "step", round(d["ms_per_step"],2), {n: round(k[n]["avg_ms"]*1000,1) for n in ("warp_residual","irls_reduce","median","gn_step")})
PY
done
