mkdir -p gpurun_out/r04t
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" "BPVO_AB_LIB=bpvo_amd/csrc/exp/libbpvo_hip_nofma.so" > gpurun_out/r04t/shard_ab.txt 2>&1; cat gpurun_out/r04t/shard_ab.txt
python -m pytest tests -m gpu -q > gpurun_out/r04t/pytest_full.txt 2>&1; tail -8 gpurun_out/r04t/pytest_full.txt
