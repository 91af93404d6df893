mkdir -p gpurun_out/r04f
python -m pytest tests -m gpu -x -q > gpurun_out/r04f/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04f/pytest_gpu.txt
tail -5 gpurun_out/r04f/pytest_gpu.txt
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 1 -- "" "lanes=1" "lanes=3" > gpurun_out/r04f/shard_ab.txt 2>&1; cat gpurun_out/r04f/shard_ab.txt
