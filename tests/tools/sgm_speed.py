"""SGM per frame for several builds of the library: python tests/tools/sgm_speed.py lib1.so lib2.so ...  (1241 x 376 x 128, 8 frames resident)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bpvo_amd import capi, synth
import __graft_entry__ as ge
n = 8
ps = [synth.make_stereo_pair(376, 1241, i) for i in range(n)]
L = np.stack([q["left"] for q in ps]); R = np.stack([q["right"] for q in ps])
torch.cuda.init()
dev = torch.device("cuda", 0)
dL, dR = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
out = torch.empty((n, 376, 1241), dtype=torch.float32, device=dev)
for lib in sys.argv[1:] or [ge.HIP_LIB]:
    hip = capi.Binding(os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib, "bpvo_hip_")
    p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(ps[0]["K"], ps[0]["b"], 376, 1241, p, n_frames=3, n_pairs=1)
    sp = ctx.default_stereo_params(128); sp.algorithm = capi.STEREO_SGM
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.stereo_bm_device(n, dL.data_ptr(), dR.data_ptr(), sp, out.data_ptr())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(os.path.basename(lib), "%.3f ms per frame" % (1e3 * dt / n), "checksum %.1f" % float(out.sum()), flush=True)
    ctx.close()
