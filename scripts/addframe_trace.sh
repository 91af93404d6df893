# kernel-trace summary of sequential addFrame 640x480 with conf/tsukuba.cfg's parameters (Intensity, CubicHermite, dense templates):
#   bash scripts/addframe_trace.sh
cd /tmp && export TMPDIR=/tmp
R=/root/repo
cat > /tmp/aft.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
import bpvo_amd
from bpvo_amd import capi, synth
hip = bpvo_amd.load()
seq = synth.make_sequence(480, 640, 25, index=21, step_rot=0.004, step_trans=0.03)
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
print("exact median: bracketed / full selections", ctx.median_path_counts(), file=sys.stderr)
PY
rm -rf /tmp/traf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/traf -- python3 /tmp/aft.py > /tmp/traf.out 2>/tmp/traf.err
python3 $R/profiles/dbstats.py /tmp/traf | head -16
python3 $R/scripts/addframe_trace_levels.py /tmp/traf ${TIMELINE_N:-0} ${TIMELINE_SKIP:-0}
grep 'exact median' /tmp/traf.err
