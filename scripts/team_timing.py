"""Per-phase timing of the team-persistent kernel under load (library built with -DBPVO_PK_TIMING as libbpvo_hip_pktiming.so):
a batch of n 1241x376 bit-planes pairs, the phases of team 0's first workgroup per pyramid level (stderr of the library)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpvo_amd
from bpvo_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = os.path.join(os.path.dirname(bpvo_amd.LIB_PATH), "libbpvo_hip_pktiming.so")
hip = capi.Binding(lib, "bpvo_hip_")
rows, cols = 376, 1241
path = f"/tmp/shard_ab_{rows}x{cols}_{max(n, 128)}.npz"
if os.path.exists(path):
    d = np.load(path); b = dict(images=d["images"][: 2 * n], disparities=d["disparities"][: 2 * n], K=d["K"], b=float(d["b"]))
else:
    b = synth.make_batch(rows, cols, n, first_index=0, workers=min(16, os.cpu_count() or 1))
p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2 * n, n_pairs=n)
for rep in range(2):
    t0 = time.perf_counter()
    poses, stats = ctx.batch_run(b["images"], b["disparities"])
    print("batch_run %d pairs (host buffers) %.2f ms; its of pair 0 %s; team launches %d" % (n, 1e3 * (time.perf_counter() - t0), stats["numIterations"][0].tolist(), ctx.team_counts()), flush=True)
