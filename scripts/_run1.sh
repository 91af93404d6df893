mkdir -p gpurun_out/r04d
rm -f gpurun_out/fuzz_outcomes.txt
python -m pytest tests -m gpu -x -q -s > gpurun_out/r04d/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r04d/pytest_gpu.txt
tail -5 gpurun_out/r04d/pytest_gpu.txt
python bench.py > gpurun_out/r04d/bench.json 2> gpurun_out/r04d/bench.err; tail -c 600 gpurun_out/r04d/bench.json
