mkdir -p gpurun_out/r04l
python -m pytest tests/test_gpu_parity.py tests/test_gpu_benched_shape.py tests/test_gpu_config5.py -m gpu -x -q -k "lazy_template or batch or lanes or host_buffer or benched or shard" > gpurun_out/r04l/pytest3.txt 2>&1; tail -6 gpurun_out/r04l/pytest3.txt
python scripts/shard_ab.py --pairs 128 --ref-pairs 1024 --steps 10 --repeat 2 -- "" "lazy_template_descriptor=0" > gpurun_out/r04l/shard_ab2.txt 2>&1; cat gpurun_out/r04l/shard_ab2.txt
