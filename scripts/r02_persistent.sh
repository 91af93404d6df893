#!/bin/bash
# persistent single-launch GN kernel: parity tests, then single-pair latency with and without it
out=gpurun_out/r02p; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q > $out/tests.log 2>&1; echo "tests rc=$?" >> $out/tests.log
tail -5 $out/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "median or linearize or estimate_pose or fused" > $out/tests2.log 2>&1; echo "tests2 rc=$?" >> $out/tests2.log
tail -3 $out/tests2.log
for on in 0 1; do
  BPVO_HIP_PERSISTENT=$on timeout 300 python scripts/b1_latency.py > $out/b1_persistent_$on.txt 2>&1
  echo "== BPVO_HIP_PERSISTENT=$on"; cat $out/b1_persistent_$on.txt
done
timeout 300 python scripts/pk_timing.py 2>&1 | tail -5 | tee $out/pk_timing.txt
