"""Sequential addFrame with the parameters of conf/perf_bitplanes.cfg / perf_intensity.cfg on a 640x480 sequence (what bench.py's
other_configs time); run under rocprofv3 --kernel-trace by scripts/addframe_timeline.sh to see the kernels of one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpvo_amd
from bpvo_amd import capi, synth

which = sys.argv[1] if len(sys.argv) > 1 else "perf_bitplanes"
hip = bpvo_amd.load()
seq = synth.make_sequence(480, 640, 14, index=3, step_rot=0.002, step_trans=0.01)
p = hip.default_params()
p.numPyramidLevels = 3; p.parameterTolerance = 1e-6; p.functionTolerance = 1e-4; p.gradientTolerance = 1e-6
p.maxIterations = 50; p.relaxTolerancesForCoarseLevels = 1; p.gradientEstimation = capi.GRAD_CD5
p.minValidDisparity = 1.0; p.goodPointThreshold = 0.75; p.verbosity = capi.VERB_SILENT
if which == "perf_bitplanes":
    p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_L2
    p.minTranslationMagToKeyFrame = 0.1; p.minRotationMagToKeyFrame = 5.0
    p.sigmaPriorToCensusTransform = 0.75; p.sigmaBitPlanes = 1.6
else:
    p.descriptor = capi.DESC_INTENSITY; p.lossFunction = capi.LOSS_HUBER; p.minSaliency = 2.5; p.nonMaxSuppRadius = 2
    p.minTranslationMagToKeyFrame = 1000.0; p.minRotationMagToKeyFrame = 1000.0; p.maxFractionOfGoodPointsToKeyFrame = 0.75
ctx = hip.create(seq["K"], seq["b"], 480, 640, p, device=0, n_frames=3, n_pairs=1)
ts = []
for img, disp in seq["frames"]:
    t0 = time.perf_counter(); r = ctx.add_frame(img, disp); ts.append((1e3 * (time.perf_counter() - t0), r["isKeyFrame"]))
    time.sleep(0.002)     # a visible gap between the calls in the trace
print(which, "addFrame ms:", " ".join(f"{t:.2f}{'K' if k else ''}" for t, k in ts))
