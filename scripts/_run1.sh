mkdir -p gpurun_out/r04s
python -m pytest tests -m gpu -q > gpurun_out/r04s/pytest_full.txt 2>&1; tail -8 gpurun_out/r04s/pytest_full.txt
python bench.py > gpurun_out/r04s/bench.json 2> gpurun_out/r04s/bench.err; tail -2 gpurun_out/r04s/bench.err; cut -c1-400 gpurun_out/r04s/bench.json
