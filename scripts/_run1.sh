mkdir -p gpurun_out/r04i
python -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py tests/test_gpu_config5.py -m gpu -x -q > gpurun_out/r04i/pytest.txt 2>&1; tail -3 gpurun_out/r04i/pytest.txt
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" > gpurun_out/r04i/shard_ab.txt 2>&1; cat gpurun_out/r04i/shard_ab.txt
for n in 16 32 64 256; do python scripts/shard_ab.py --pairs $n --steps 10 --repeat 1 -- "" "team=0" >> gpurun_out/r04i/sweep.txt 2>&1; done; cat gpurun_out/r04i/sweep.txt
