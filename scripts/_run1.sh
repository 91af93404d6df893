mkdir -p gpurun_out/r04p
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" "BPVO_AB_LIB=bpvo_amd/csrc/exp/libbpvo_hip_w4.so" > gpurun_out/r04p/shard_ab.txt 2>&1; cat gpurun_out/r04p/shard_ab.txt
