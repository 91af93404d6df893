"""Soak: many random synthetic pairs through the HIP batch path and the CPU oracle; reports the worst pose disagreement and
any difference in per-level status.  usage: soak_parity.py [n_pairs] [rows cols] [descriptor] [loss]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
import bpvo_amd
from bpvo_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (376, 1241)
descriptor = sys.argv[4] if len(sys.argv) > 4 else "bitplanes"
loss = sys.argv[5] if len(sys.argv) > 5 else "tukey"
hip = bpvo_amd.load()
orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
b = synth.make_batch(rows, cols, n, first_index=5000, workers=min(16, os.cpu_count() or 1))
res = {}
for name, bind in (("hip", hip), ("orc", orc)):
    p = bind.default_params(); p.numPyramidLevels = 4
    p.descriptor = capi.DESC_BITPLANES if descriptor == "bitplanes" else capi.DESC_INTENSITY
    p.lossFunction = {"tukey": capi.LOSS_TUKEY, "huber": capi.LOSS_HUBER, "l2": capi.LOSS_L2}[loss]
    p.verbosity = capi.VERB_SILENT
    ctx = bind.create(b["K"], b["b"], rows, cols, p, n_frames=2 * n, n_pairs=n)
    if name == "orc":
        ctx.call("set_num_threads", 8)
    t0 = time.perf_counter()
    res[name] = ctx.batch_run(b["images"], b["disparities"])
    print(name, "%.1f s" % (time.perf_counter() - t0))
    ctx.close()
(ph, sh), (po, so) = res["hip"], res["orc"]
E = np.einsum("nji,njk->nik", po[:, :3, :3].astype(np.float64), ph[:, :3, :3].astype(np.float64))
w = 0.5 * np.stack([E[:, 2, 1] - E[:, 1, 2], E[:, 0, 2] - E[:, 2, 0], E[:, 1, 0] - E[:, 0, 1]], axis=1)
rot = np.arcsin(np.minimum(1.0, np.linalg.norm(w, axis=1)))
tr = np.linalg.norm(po[:, :3, 3].astype(np.float64) - ph[:, :3, 3].astype(np.float64), axis=1)
print("pairs %d  %dx%d %s %s" % (n, cols, rows, descriptor, loss))
print("rotation  : rmse %.3e  max %.3e rad (bar 1e-4)" % (np.sqrt(np.mean(rot ** 2)), rot.max()))
print("translation: rmse %.3e  max %.3e m   (bar 1e-3)" % (np.sqrt(np.mean(tr ** 2)), tr.max()))
print("iterations differ in %d of %d (pair, level) cells; status differs in %d" % (
    int((sh["numIterations"] != so["numIterations"]).sum()), sh["numIterations"].size, int((sh["status"] != so["status"]).sum())))
for k in range(min(6, n)):
    print("pair", k, "hip its", sh["numIterations"][k].tolist(), "status", [hex(x) for x in sh["status"][k].tolist()],
          "| orc its", so["numIterations"][k].tolist(), "status", [hex(x) for x in so["status"][k].tolist()],
          "| fE hip", np.round(sh["finalError"][k], 4).tolist(), "orc", np.round(so["finalError"][k], 4).tolist())
print("worst pairs:", np.argsort(-tr)[:5].tolist(), tr[np.argsort(-tr)[:5]].tolist())
sys.exit(0 if rot.max() <= 1e-4 and tr.max() <= 1e-3 else 1)
