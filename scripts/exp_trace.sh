# like exp_variants.sh but reports rocprofv3 kernel averages for the frame kernels at 256 pairs
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  echo "== $v"; bash scripts/trace_kernels.sh 256 | grep -E "blur|census|saliency|select_|pyrdown" 
done
cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so
