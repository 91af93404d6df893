#!/bin/bash
# Stereo front-end evidence (run through gpurun from the repo root): throughput lines of scripts/stereo_bench.py and the rocprofv3 kernel
# trace of the same script -> gpurun_out/profiles_<tag>/<tag>_stereo.txt
set -u
TAG=${1:-r03}
R=$(pwd)
mkdir -p "$R/gpurun_out/profiles_$TAG"
OUT="$R/gpurun_out/profiles_$TAG/${TAG}_stereo.txt"
python3 scripts/stereo_bench.py 8 > "$OUT" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/st_prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_prof -- python3 "$R/scripts/stereo_bench.py" 4 > /dev/null 2>&1
cd "$R"
python3 - >> "$OUT" <<PY
import glob, sqlite3
fs = sorted(glob.glob("/tmp/st_prof/*/*_results.db"))
db = sqlite3.connect(fs[-1])
print()
print("rocprofv3 --kernel-trace --stats -- python3 scripts/stereo_bench.py 4   (4 pairs per batch, 2 + 5 calls per configuration; kernel, calls, total us, average us, percent)")
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 18"):
    print("%-100s %6d %12.1f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))
PY
cat "$OUT" | cut -c1-250
