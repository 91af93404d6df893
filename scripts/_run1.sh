mkdir -p gpurun_out/r04e
python -m pytest tests/test_gpu_parity.py -x -q -k "latch or unsupported" > gpurun_out/r04e/pytest_latch.txt 2>&1; tail -30 gpurun_out/r04e/pytest_latch.txt
