mkdir -p gpurun_out/r04h
python tests/tools/sgm_bisect.py bpvo_amd/csrc/libbpvo_hip.so bpvo_amd/csrc/exp/libbpvo_hip_pf8.so > gpurun_out/r04h/bisect4.txt 2>&1; cat gpurun_out/r04h/bisect4.txt
python tests/tools/sgm_speed.py bpvo_amd/csrc/libbpvo_hip.so bpvo_amd/csrc/exp/libbpvo_hip_pf8.so bpvo_amd/csrc/exp/libbpvo_hip_pf32.so > gpurun_out/r04h/speed4.txt 2>&1; cat gpurun_out/r04h/speed4.txt
