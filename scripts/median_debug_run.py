import sys, os, numpy as np
R="/root/repo"; sys.path.insert(0, R)
from bpvo_amd import capi, synth
hip = capi.Binding(os.path.join(R,"bpvo_amd","csrc","libbpvo_hip_dbg.so"), "bpvo_hip_")
seq = synth.make_sequence(480, 640, 4, index=21, step_rot=0.004, step_trans=0.03)
p = hip.default_params()
p.numPyramidLevels = 3; p.parameterTolerance = 1e-6; p.functionTolerance = 1e-4; p.gradientTolerance = 1e-6
p.maxIterations = 55; p.relaxTolerancesForCoarseLevels = 0; p.gradientEstimation = capi.GRAD_CD5
p.minValidDisparity = 1.0; p.goodPointThreshold = 0.75; p.verbosity = capi.VERB_SILENT
p.descriptor = capi.DESC_INTENSITY; p.lossFunction = capi.LOSS_HUBER; p.minSaliency = 0.001; p.nonMaxSuppRadius = 0
p.minTranslationMagToKeyFrame = 0.05; p.minRotationMagToKeyFrame = 2.5; p.maxFractionOfGoodPointsToKeyFrame = 0.5
p.interp = capi.INTERP_CUBIC_HERMITE
ctx = hip.create(seq["K"], seq["b"], 480, 640, p, device=0, n_frames=3, n_pairs=1)
for img, disp in seq["frames"]:
    ctx.add_frame(img, disp)
