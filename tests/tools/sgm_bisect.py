"""Which stage of the SGM device path differs from the oracle: the bit-exactness case of tests/test_stereo.py against several builds of
the library (scripts/build_exp.sh variants that switch single kernels back).  python tests/tools/sgm_bisect.py lib1.so lib2.so ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from bpvo_amd import capi, synth
import test_stereo as ts

orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
cases = [(376, 1241, dict(ndisp=128)), (64, 300, dict(ndisp=256, factor=64.0)), (97, 203, dict(ndisp=48, cap=40, thr=2, cw=0.5))]
for lib in sys.argv[1:] or [ge.HIP_LIB]:
    hip = capi.Binding(os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib, "bpvo_hip_")
    for rows, cols, kw in cases:
        d = synth.make_stereo_pair(rows, cols, 4, z0=8.0 if cols > 700 else 4.0)
        rng = np.random.default_rng(cols)
        left, right = d["left"].copy(), d["right"].copy()
        right[: rows // 4] = rng.integers(0, 256, (rows // 4, cols), dtype=np.uint8)
        left[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
        right[rows // 2: rows // 2 + 20, 30: 30 + cols // 4] = 100
        p = hip.default_params(); p.numPyramidLevels = 2; p.verbosity = capi.VERB_SILENT
        ctx = hip.create(d["K"], d["b"], rows, cols, p, n_frames=3, n_pairs=1)
        got = ctx.stereo_bm(left, right, ts._sgm_params(ctx, **kw))
        want = ts.orc_sgm(orc, left, right, **kw)
        bad = np.argwhere(got != want)
        print(os.path.basename(lib), (rows, cols, kw), "differing pixels:", len(bad), "got==0:", int((got[got != want] == 0).sum()), "want==0:", int((want[got != want] == 0).sum()),
              bad[:4].tolist(), flush=True)
        ctx.close()
