mkdir -p gpurun_out/r04l
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lazy_template or batch or lanes or host_buffer" > gpurun_out/r04l/pytest.txt 2>&1; tail -15 gpurun_out/r04l/pytest.txt
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" "lazy_template_descriptor=0" > gpurun_out/r04l/shard_ab.txt 2>&1; cat gpurun_out/r04l/shard_ab.txt
