# experimental builds for scripts/exp_value.sh: bash scripts/build_exp.sh name "-DMACRO=.. -D.." [name2 "flags2" ...]
mkdir -p bpvo_amd/csrc/exp
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (cd bpvo_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wno-unused-function $flags -o exp/libbpvo_hip_$name.so bpvo_hip.hip kernels_frame.hip kernels_gn.hip kernels_stereo.hip kernels_sgm.hip) && echo "built $name ($flags)" &
done
wait
