mkdir -p gpurun_out/r04c
for n in 128 1024; do python scripts/shard_phases.py --pairs $n; BPVO_HIP_LANES=1 python scripts/shard_phases.py --pairs $n --steps 4; done > gpurun_out/r04c/phases.txt 2>&1
cat gpurun_out/r04c/phases.txt
