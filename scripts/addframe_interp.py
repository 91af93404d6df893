import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import bpvo_amd
from bpvo_amd import capi, synth
hip = bpvo_amd.load()
seq = synth.make_sequence(480, 640, 25, index=21, step_rot=0.004, step_trans=0.03)
def run(interp, desc):
    p = hip.default_params()
    p.numPyramidLevels = 3; p.parameterTolerance = 1e-6; p.functionTolerance = 1e-4; p.gradientTolerance = 1e-6
    p.maxIterations = 55; p.relaxTolerancesForCoarseLevels = 0; p.gradientEstimation = capi.GRAD_CD5
    p.minValidDisparity = 1.0; p.goodPointThreshold = 0.75; p.verbosity = capi.VERB_SILENT
    p.descriptor = desc; p.lossFunction = capi.LOSS_HUBER; p.minSaliency = 0.001; p.nonMaxSuppRadius = 0
    p.sigmaPriorToCensusTransform = 0.75; p.sigmaBitPlanes = 1.75
    p.minTranslationMagToKeyFrame = 0.05; p.minRotationMagToKeyFrame = 2.5; p.maxFractionOfGoodPointsToKeyFrame = 0.5
    p.interp = interp
    ctx = hip.create(seq["K"], seq["b"], 480, 640, p, device=0, n_frames=3, n_pairs=1)
    frames = seq["frames"]
    ctx.add_frame(*frames[0])
    t0 = time.perf_counter(); nk = 0; its = 0
    for img, disp in frames[1:]:
        r = ctx.add_frame(img, disp); nk += int(r["isKeyFrame"]); its += sum(s["numIterations"] for s in r["stats"])
    dt = time.perf_counter() - t0
    lv, gu = ctx.persistent_counts()
    ctx.close()
    return 1e3 * dt / (len(frames) - 1), nk, its / (len(frames) - 1), lv
for desc, dn in ((capi.DESC_INTENSITY, "intensity"), (capi.DESC_BITPLANES, "bitplanes")):
    for interp, name in ((0, "kLinear"), (3, "kCubicHermite"), (2, "kCubic"), (1, "kCosine")):
        ms, nk, its, lv = run(interp, desc)
        print(f"addFrame 640x480 conf/tsukuba.cfg parameters, {dn}, {name}: {ms:.2f} ms per frame, {nk} key frames, {its:.0f} iterations per frame, persistent levels {lv}", flush=True)
