# per-stage timing of the frame/template kernels (HIP events of bench.py) at the default batch
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 ${@} 2>&1 | tail -1 > /tmp/sb.json; python - <<PY
import json
d=json.load(open("/tmp/sb.json"))
k=d["kernels"]
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"],2))
for n in ("pyramid","descriptor","saliency_select","normalization","template_build","warp_residual"):
    print("  ", n, "ms/step", round(k[n]["total_ms"]/d["steps"],2), "alg GB/s", round(k[n]["algorithmic_GBps"]))
PY
