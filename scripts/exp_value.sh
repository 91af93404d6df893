# A/B of experimental builds (bpvo_amd/csrc/exp/libbpvo_hip_<name>.so) against the current build: headline value and K6 GB/s
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
for rep in 1 2; do
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --input-cache /tmp/bpvo_bench_inputs 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', round(d['value']), round(d['ms_per_step'],2), round(r['achieved']), round(r['avg_launch_ms']*1000,1))"
done; done
cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so
