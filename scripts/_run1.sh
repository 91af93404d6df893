mkdir -p gpurun_out/r04u
bash profiles/collect_profiles.sh r04 > gpurun_out/r04u/collect.log 2>&1; tail -4 gpurun_out/r04u/collect.log | cut -c1-300
python bench.py > gpurun_out/r04u/bench.json 2> gpurun_out/r04u/bench.err; tail -2 gpurun_out/r04u/bench.err; cut -c1-300 gpurun_out/r04u/bench.json
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04u/smoke.txt 2>&1; tail -3 gpurun_out/r04u/smoke.txt
