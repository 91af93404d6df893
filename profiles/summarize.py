#!/usr/bin/env python3
"""Turns the rocprofv3 databases written by collect_profiles.sh into small text/JSON summaries.

usage: summarize.py <gpurun_out/profiles_TAG> <TAG>   -> writes <dir>/<TAG>_kernel_trace_stats.txt,
       <TAG>_pmc_summary.txt and <TAG>_traffic.json (copy them into profiles/ to commit)."""
import glob
import json
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
out = []


def db_of(sub):
    # newest database: gpurun merges every run's output into the same local directory
    fs = sorted(glob.glob(os.path.join(src, sub, "*", "*_results.db")), key=os.path.getmtime)
    return sqlite3.connect(fs[-1]) if fs else None


for sub, suffix, env in (("trace", "", ""), ("trace1", "_1lane", "BPVO_HIP_OPTIONS=lanes=1 ")):
    db = db_of(sub)
    if not db:
        continue
    rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    lines = [env + "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-pairs 0 --other-configs 0 --gen-workers 1 --input-cache /tmp/bpvo_bench_inputs   (durations in us)",
             "%-100s %8s %16s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "%")]
    for r in rows:
        lines.append("%-100s %8d %16.1f %12.2f %7.2f" % (r[0][:100], r[1], r[2], r[3], r[4]))
    try:
        bench = json.loads([l for l in open(os.path.join(src, sub + "_bench.json")) if l.startswith("{")][-1])
        lines.append("")
        lines.append("bench.py line of the same run: value=%.1f %s, ms_per_step=%.3f, roofline=%s, roofline_timed_region=%s" %
                     (bench["value"], bench["unit"], bench["ms_per_step"], json.dumps(bench["roofline"]), json.dumps(bench.get("roofline_timed_region"))))
    except Exception as e:  # noqa: BLE001
        lines.append("(no bench json: %s)" % e)
    open(os.path.join(src, f"{tag}_kernel_trace_stats{suffix}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:10]))

per = {}
for sub in sorted(glob.glob(os.path.join(src, "pmc*"))):
    if not os.path.isdir(sub):
        continue
    d = db_of(os.path.basename(sub))
    if not d:
        continue
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration), avg(grid_size) from counters_collection "
         "group by kernel_name, counter_name")
    for k, c, n, v, dur, grid in d.execute(q):
        if "bpvo_hip" not in k:
            continue
        per.setdefault(k, {})[c] = dict(launches=n, avg=v, avg_duration_ns=dur)
lines = ["PMC averages per launch on the benched workload: BPVO_HIP_OPTIONS=lanes=1 bench.py --steps 2 --warmup 0 (1024 pairs, converge mode, AlgorithmParameters()",
         "tolerances; the launches shrink as pairs converge: averages are over all launches).  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them."]
for k in sorted(per):
    lines.append("")
    lines.append(k[:120])
    for c in sorted(per[k]):
        e = per[k][c]
        lines.append("    %-40s %14.6g   (n=%d, avg kernel %.1f us)" % (c, e["avg"], e["launches"], (e["avg_duration_ns"] or 0) / 1e3))
open(os.path.join(src, f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")

# HBM traffic of warp_residual per point (exact request sizes) for bench.py's roofline.traffic
traffic = {}
for k, cs in per.items():
    if "warp_residual_kernel<8" in k and "TCC_EA0_RDREQ_128B_sum" in cs:
        rd = 128 * cs["TCC_EA0_RDREQ_128B_sum"]["avg"] + 64 * cs["TCC_EA0_RDREQ_64B_sum"]["avg"] + 32 * cs["TCC_EA0_RDREQ_32B_sum"]["avg"]
        wr64 = cs.get("TCC_EA0_WRREQ_64B_sum", {}).get("avg", 0.0)
        wr = cs.get("TCC_EA0_WRREQ_sum", {}).get("avg", 0.0)
        wbytes = 64 * wr64 + 32 * max(0.0, wr - wr64)
        traffic["warp_residual_read_bytes_per_launch"] = rd
        traffic["warp_residual_write_bytes_per_launch"] = wbytes
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            traffic["guide_corrected_bytes_per_launch"] = 1024.0 * (2.0 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"])
            traffic["fetch_size_kib_raw"] = cs["FETCH_SIZE"]["avg"]
            traffic["write_size_kib_raw"] = cs["WRITE_SIZE"]["avg"]
try:
    pj = json.loads([l for l in open(os.path.join(src, "pmc3.json")) if l.startswith("{")][-1])
    k6 = [k for k in per if "warp_residual_kernel<8" in k][0]
    launches = per[k6]["TCC_EA0_RDREQ_128B_sum"]["launches"]
    # points warp_residual itself processed (linearisations with a frozen scale go through irls_reduce's fused path)
    fp = pj.get("fused_path") or {"points": 0, "of": pj["points_linearized_rank0"]}
    pts_per_launch = (fp["of"] - fp["points"]) / launches
    traffic["fused_fraction_in_pmc_run"] = fp["points"] / max(1, fp["of"])
    traffic["pmc_config"] = pj["config"]["workload"]
    traffic["points_per_launch"] = pts_per_launch
    traffic["warp_residual_hbm_bytes_per_point"] = (traffic["warp_residual_read_bytes_per_launch"]
                                                    + traffic["warp_residual_write_bytes_per_launch"]) / pts_per_launch
    if "guide_corrected_bytes_per_launch" in traffic:
        traffic["guide_corrected_bytes_per_point"] = traffic["guide_corrected_bytes_per_launch"] / pts_per_launch
    traffic["algorithmic_bytes_per_point"] = 210
except Exception as e:  # noqa: BLE001
    traffic["error"] = str(e)
# ... and of irls_reduce (one launch serves the plain and the fused path): bytes per linearised point, for bench.py's second roofline block
try:
    k8 = [k for k in per if "irls_reduce_both_kernel" in k and "TCC_EA0_RDREQ_128B_sum" in per[k]][0]
    cs = per[k8]
    rd = 128 * cs["TCC_EA0_RDREQ_128B_sum"]["avg"] + 64 * cs["TCC_EA0_RDREQ_64B_sum"]["avg"] + 32 * cs["TCC_EA0_RDREQ_32B_sum"]["avg"]
    wr64 = cs.get("TCC_EA0_WRREQ_64B_sum", {}).get("avg", 0.0)
    wr = cs.get("TCC_EA0_WRREQ_sum", {}).get("avg", 0.0)
    wbytes = 64 * wr64 + 32 * max(0.0, wr - wr64)
    pj = json.loads([l for l in open(os.path.join(src, "pmc3.json")) if l.startswith("{")][-1])
    ppl = pj["points_linearized_rank0"] / cs["TCC_EA0_RDREQ_128B_sum"]["launches"]
    traffic["irls_reduce_read_bytes_per_launch"] = rd
    traffic["irls_reduce_write_bytes_per_launch"] = wbytes
    traffic["irls_reduce_points_per_launch"] = ppl
    traffic["irls_reduce_hbm_bytes_per_point"] = (rd + wbytes) / ppl
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        traffic["irls_reduce_guide_corrected_bytes_per_point"] = 1024.0 * (2.0 * cs["FETCH_SIZE"]["avg"] + cs["WRITE_SIZE"]["avg"]) / ppl
except Exception as e:  # noqa: BLE001
    traffic["irls_reduce_error"] = str(e)
# the same for the timing-tolerance batch (tpmc1 / tpmc2)
tper = {}
for sub in sorted(glob.glob(os.path.join(src, "tpmc*"))):
    if os.path.isdir(sub):
        d = db_of(os.path.basename(sub))
        if d:
            for k, c, n, v, dur, grid in d.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), avg(grid_size) from counters_collection group by kernel_name, counter_name"):
                if "warp_residual_kernel<8" in k:
                    tper[c] = dict(launches=n, avg=v)
try:
    tj = json.loads([l for l in open(os.path.join(src, "tpmc1.json")) if l.startswith("{")][-1])
    rd = 128 * tper["TCC_EA0_RDREQ_128B_sum"]["avg"] + 64 * tper["TCC_EA0_RDREQ_64B_sum"]["avg"] + 32 * tper["TCC_EA0_RDREQ_32B_sum"]["avg"]
    wr64 = tper["TCC_EA0_WRREQ_64B_sum"]["avg"]; wr = tper["TCC_EA0_WRREQ_sum"]["avg"]
    fp = tj["fused_path"]
    ppl = (fp["of"] - fp["points"]) / tper["TCC_EA0_RDREQ_128B_sum"]["launches"]
    traffic["timing_tolerance_batch"] = {"config": tj["config"]["workload"], "points_per_launch": ppl,
                                         "warp_residual_hbm_bytes_per_point": (rd + 64 * wr64 + 32 * max(0.0, wr - wr64)) / ppl,
                                         "tap_cache": tj.get("tap_cache")}
except Exception as e:  # noqa: BLE001
    traffic["timing_tolerance_batch"] = {"error": str(e)}
open(os.path.join(src, f"{tag}_traffic.json"), "w").write(json.dumps(traffic, indent=1) + "\n")
print(json.dumps(traffic))
