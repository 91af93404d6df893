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
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 30"):
    print("%-100s %6d %12.1f %10.2f %6.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))
PY
cat "$OUT" | cut -c1-250

# ---- PMC passes over the SGM kernels (own runs: --pmc with --kernel-trace only), same script, 2 pairs per batch
PMC_OUT="$R/gpurun_out/profiles_$TAG/${TAG}_stereo_pmc.txt"
: > "$PMC_OUT"
cd /tmp
i=0
for CNT in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/st_pmc$i
  timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d /tmp/st_pmc$i -- python3 "$R/scripts/stereo_bench.py" 2 > /dev/null 2>&1
  echo "pmc pass $i rc=$?"
done
cd "$R"
python3 - >> "$PMC_OUT" <<PY
import glob, os, sqlite3
per = {}
for i in (1, 2, 3):
    fs = sorted(glob.glob("/tmp/st_pmc%d/*/*_results.db" % i), key=os.path.getmtime)
    if not fs:
        continue
    db = sqlite3.connect(fs[-1])
    q = "select kernel_name, counter_name, count(*), avg(value), avg(duration), max(grid_size) from counters_collection where kernel_name like '%sgm_%' or kernel_name like '%sgbm_%' or kernel_name like '%stereo_%' group by kernel_name, counter_name, grid_size"
    for k, c, n, v, dur, grid in db.execute(q):
        import re
        m = re.search(r"(sgbm_\w+|sgm_\w+|stereo_\w+)(<[^>]*>)?", k)
        per.setdefault((m.group(0) if m else k[:40], grid), {})[c] = (v, n, dur)
print("PMC averages per launch of the stereo kernels, scripts/stereo_bench.py 2 (1241x376 / 128 and 640x480 / 64 disparities), one rocprofv3 --pmc pass per counter set.")
print("HBM bytes from the request-size counters: read = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B, written = 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B).")
for (k, grid), cs in sorted(per.items(), key=lambda kv: -(kv[1].get("SQ_WAVE_CYCLES", (0, 0, 0))[2] or 0)):
    dur = [v[2] for v in cs.values() if v[2]]
    us = (sum(dur) / len(dur)) / 1e3 if dur else 0.0
    line = "%-34s grid %-9d %9.1f us" % (k[:34], grid, us)
    if "TCC_EA0_RDREQ_128B_sum" in cs:
        rd = 32 * cs["TCC_EA0_RDREQ_32B_sum"][0] + 64 * cs["TCC_EA0_RDREQ_64B_sum"][0] + 128 * cs["TCC_EA0_RDREQ_128B_sum"][0]
        line += "  read %8.1f MB" % (rd / 1e6)
        if "TCC_EA0_WRREQ_sum" in cs:
            wr = 64 * cs["TCC_EA0_WRREQ_64B_sum"][0] + 32 * max(0.0, cs["TCC_EA0_WRREQ_sum"][0] - cs["TCC_EA0_WRREQ_64B_sum"][0])
            line += " written %8.1f MB -> %6.0f GB/s of HBM (%.2f of 8 TB/s)" % (wr / 1e6, (rd + wr) / (us * 1e-6) / 1e9 if us else 0, (rd + wr) / (us * 1e-6) / 8e12 if us else 0)
            hit, miss = cs.get("TCC_HIT_sum", (0,))[0], cs.get("TCC_MISS_sum", (0,))[0]
            if hit + miss > 0:
                line += "  L2 hit %.2f" % (hit / (hit + miss))
    if "SQ_WAVE_CYCLES" in cs:
        wc = cs["SQ_WAVE_CYCLES"][0]
        line += "  | waves waiting %.2f, issue-stalled %.2f, VALU active %.2f of wave cycles; %.3g VALU instructions per launch" % (
            cs["SQ_WAIT_ANY"][0] / wc, cs["SQ_WAIT_INST_ANY"][0] / wc, cs["SQ_ACTIVE_INST_VALU"][0] / wc, cs["SQ_INSTS_VALU"][0])
    print(line)
PY
cat "$PMC_OUT" | cut -c1-260
