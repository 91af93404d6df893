# swaps experimental builds of libbpvo_hip.so (bpvo_amd/csrc/exp/) in and prints the warp_residual timing of each
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --fixed-iters 20 --profile-all 2>&1 | tail -1 > /tmp/ev.json; python - <<PY
import json
d=json.load(open("/tmp/ev.json"))
k=d["kernels"]
print("$v", "step", round(d["ms_per_step"],2), {n: (round(k[n]["avg_ms"]*1000,1), round(k[n]["units_per_launch"])) for n in ("warp_residual","irls_reduce","median","gn_step")})
PY
done
cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so
