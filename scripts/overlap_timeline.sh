# (needs profiles/r06_team_overlap_rejected.diff applied for OPT=1)  timeline of one bpvo_hip_batch_run of P pairs with option team_overlap_finest_level = $2 (default 1): per stream the kernels with start / end
# relative to the step's first kernel.   bash scripts/overlap_timeline.sh [pairs] [0|1]
P=${1:-128}; OPT=${2:-1}
cd /tmp && export TMPDIR=/tmp
R=/root/repo
cat > /tmp/ovl_run.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from bpvo_amd import capi, synth
import bpvo_amd, torch
n = $P
import os
path = "/tmp/ovl_batch_%d.npz" % n
if os.path.exists(path):
    d = np.load(path); batch = dict(images=d["images"], disparities=d["disparities"], K=d["K"], b=float(d["b"]))
else:
    batch = synth.make_batch(376, 1241, n, first_index=1000, workers=1 if "--nofork" in sys.argv else 8)
    np.savez(path, images=batch["images"], disparities=batch["disparities"], K=batch["K"], b=batch["b"])
if "--gen" in sys.argv: sys.exit(0)
torch.cuda.init(); dev = torch.device("cuda", 0)
hip = bpvo_amd.load()
p = hip.default_params(); p.numPyramidLevels = 4; p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_TUKEY; p.verbosity = capi.VERB_SILENT
ctx = hip.create(batch["K"], batch["b"], 376, 1241, p, device=0, n_frames=2 * n, n_pairs=n)
ctx.set_option("team_overlap_finest_level", $OPT)
d_i, d_d = torch.from_numpy(batch["images"]).to(dev), torch.from_numpy(batch["disparities"]).to(dev)
for _ in range(3): ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
torch.cuda.synchronize()
PY
python3 /tmp/ovl_run.py --gen
rm -rf /tmp/tro; timeout 300 rocprofv3 --kernel-trace -d /tmp/tro -- python3 /tmp/ovl_run.py > /tmp/tro.out 2>/tmp/tro.err
tail -2 /tmp/tro.err
python3 - <<PY
import glob, sqlite3, os, collections
fs = sorted(glob.glob("/tmp/tro/*/*_results.db"), key=os.path.getmtime)
db = sqlite3.connect(fs[-1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
rows = list(db.execute("select name, start, end, %s from kernels order by start" % qcol))
short = lambda n: n.split("(")[0].split("::")[-1].split("<")[0].replace("_kernel", "")
ing = [i for i, r in enumerate(rows) if "ingest" in r[0]]
rows = rows[ing[-1]:]
t0 = rows[0][1]
print("option team_overlap_finest_level = $OPT, $P pairs: last step, %d kernels, %.2f ms" % (len(rows), (max(r[2] for r in rows) - t0) / 1e6))
for n, s, e, q in rows:
    print("  q%-3s %-28s %8.3f -> %8.3f ms  (%7.3f)" % (q, short(n)[:28], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
