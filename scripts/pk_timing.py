"""Per-phase timing of the persistent GN kernel (library built with -DBPVO_PK_TIMING as bpvo_amd/csrc/libbpvo_hip_pktiming.so):
one 1241x376 bit-planes pair, estimate_pose; the library prints the per-level averages on stderr (a build with -DBPVO_PK_TIMING: scripts/build_exp.sh pkt "-DBPVO_PK_TIMING", BPVO_AB_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpvo_amd
from bpvo_amd import capi, synth

lib = os.path.join(os.path.dirname(bpvo_amd.LIB_PATH), "libbpvo_hip_pktiming.so")
hip = capi.Binding(lib, "bpvo_hip_")
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (376, 1241)
desc = sys.argv[3] if len(sys.argv) > 3 else "bitplanes"
b = synth.make_batch(rows, cols, 1, first_index=0, workers=1)
p = hip.default_params(); p.numPyramidLevels = 4
p.descriptor = capi.DESC_BITPLANES if desc == "bitplanes" else capi.DESC_INTENSITY
p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
ctx.frame_set_data(0, b["images"][0], b["disparities"][0]); ctx.frame_set_template(0); ctx.frame_set_data(1, b["images"][1], b["disparities"][1])
for rep in range(3):
    t0 = time.perf_counter()
    T, st = ctx.estimate_pose(0, 0, 1)
    dt = time.perf_counter() - t0
    print("estimate_pose %.2f ms, iterations %s, points %s" % (1e3 * dt, [s["numIterations"] for s in st], [ctx.num_points(0, l) for l in range(4)]), flush=True)
