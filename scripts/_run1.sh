mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "step_taken or lanes or batch" > gpurun_out/r04n/pytest1.txt 2>&1; tail -6 gpurun_out/r04n/pytest1.txt
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" "step_in_reduce=0" "BPVO_AB_LIB=bpvo_amd/csrc/exp/libbpvo_hip_ladder.so,step_in_reduce=0" > gpurun_out/r04n/shard_ab.txt 2>&1; cat gpurun_out/r04n/shard_ab.txt
python scripts/shard_ab.py --pairs 256 --ref-pairs 512 --steps 10 --repeat 2 -- "" "step_in_reduce=0" > gpurun_out/r04n/shard_ab256.txt 2>&1; cat gpurun_out/r04n/shard_ab256.txt
