bash profiles/collect_profiles.sh r04 > gpurun_out/r04_collect.log 2>&1; tail -5 gpurun_out/r04_collect.log
bash profiles/collect_stereo.sh r04 > gpurun_out/r04_collect_stereo.log 2>&1; tail -3 gpurun_out/r04_collect_stereo.log
