# the frame stage launch by launch (scripts/frame_stage_trace.sh) for the current build and experimental builds (scripts/build_exp.sh):
# bash scripts/frame_stage_ab.sh "<grep pattern of the kernels to show>" name1 name2 ...
# The variants are installed over the product library one after the other; whatever happens (an error, the timeout of the trace script,
# Ctrl-C) the EXIT trap puts the product build back.
set -e
PAT=$1; shift
for v in "$@"; do [ -f bpvo_amd/csrc/exp/libbpvo_hip_$v.so ] || { echo "no such build: $v (scripts/build_exp.sh)"; exit 1; }; done
cp bpvo_amd/csrc/libbpvo_hip.so /tmp/libbpvo_hip.base.so
trap 'cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so' EXIT
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/libbpvo_hip.base.so bpvo_amd/csrc/libbpvo_hip.so; else cp bpvo_amd/csrc/exp/libbpvo_hip_$v.so bpvo_amd/csrc/libbpvo_hip.so; fi
  echo "== $v"
  bash scripts/frame_stage_trace.sh 1024 2>&1 | grep "$PAT\|frame stage" || true
done
