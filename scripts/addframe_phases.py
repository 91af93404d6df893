"""Where a sequential addFrame spends its time: the C-ABI stages it is made of, timed one by one on the same frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpvo_amd
from bpvo_amd import capi, synth

hip = bpvo_amd.load()


def timed(f, reps=20):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    return 1e3 * (time.perf_counter() - t0) / reps, out


for rows, cols, desc, loss in ((480, 640, capi.DESC_INTENSITY, capi.LOSS_HUBER), (480, 640, capi.DESC_BITPLANES, capi.LOSS_TUKEY),
                               (376, 1241, capi.DESC_BITPLANES, capi.LOSS_TUKEY)):
    seq = synth.make_sequence(rows, cols, 6, index=3, step_rot=0.002, step_trans=0.01)
    p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = desc; p.lossFunction = loss; p.verbosity = capi.VERB_SILENT
    ctx = hip.create(seq["K"], seq["b"], rows, cols, p, n_frames=3, n_pairs=1)
    (i0, d0), (i1, d1) = seq["frames"][0], seq["frames"][1]
    t_set, _ = timed(lambda: ctx.frame_set_data(0, i0, d0))
    t_tmpl, _ = timed(lambda: ctx.frame_set_template(0))
    ctx.frame_set_data(1, i1, d1)
    t_est, (T, st) = timed(lambda: ctx.estimate_pose(0, 0, 1))
    t_frac, _ = timed(lambda: ctx.fraction_good(0, 0.75))
    its = sum(s["numIterations"] for s in st)
    vo = hip.create(seq["K"], seq["b"], rows, cols, p, n_frames=3, n_pairs=1)
    ts = []
    for img, disp in seq["frames"]:
        t0 = time.perf_counter(); r = vo.add_frame(img, disp); ts.append((1e3 * (time.perf_counter() - t0), r["isKeyFrame"]))
    print(f"{cols}x{rows} desc {desc:#x}: setData {t_set:.3f} ms, setTemplate {t_tmpl:.3f} ms, estimatePose {t_est:.3f} ms ({its} it, "
          f"{1e3 * t_est / max(1, its):.1f} us/it), fractionGood {t_frac:.3f} ms; addFrame " + " ".join(f"{t:.2f}{'K' if k else ''}" for t, k in ts))
