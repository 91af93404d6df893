# the frame stage launch by launch (scripts/frame_stage_trace.sh) for the current build and experimental builds (scripts/build_exp.sh):
# bash scripts/frame_stage_ab.sh "<grep pattern of the kernels to show>" name1 name2 ...
PAT=$1; shift
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  echo "== $v"
  bash scripts/frame_stage_trace.sh 1024 2>&1 | grep "$PAT\|frame stage"
done
cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so
