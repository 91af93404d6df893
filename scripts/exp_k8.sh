# like exp_value.sh, but also prints the irls_reduce / median / gn_step averages (--profile-all)
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --profile-all --input-cache /tmp/bpvo_bench_inputs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$v', round(d['value']), {n: round(k[n]['avg_ms']*1000,1) for n in ('warp_residual','irls_reduce','median','gn_step')})"
done
cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so
