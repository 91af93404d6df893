#!/usr/bin/env python3
"""The reference's OWN spread under its summation orders (CPU only, no GPU).

LinearSystemBuilder::Run sums the normal equations serially in the default build and with tbb::parallel_reduce when WITH_TBB is on
(bpvo/linear_system_builder.cc:91-131,233-237): the decomposition — and with it the rounding of H, G and f_norm — is whatever the
TBB partitioner picks.  The oracle restates that as contiguous chunks summed in chunk order (`set_num_threads(n)`: n chunks) and
also offers an f64 accumulation (`set_reduction(1)`, a test instrument).  This script runs picked pairs of the config-5 shard under
each of those orders and prints, per pair, the per-level iteration counts / statuses of every variant and how far the final poses
of the variants lie from the single-threaded one: that spread is the envelope the GPU result is judged against
(tests/test_gpu_config5.py).

usage: oracle_envelope.py [--pairs 80,0,4] [--tolerances timing|default] [--threads 1,2,3,4,5,6,7,8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi, synth  # noqa: E402
from util import make_params, pose_error  # noqa: E402

ROWS, COLS, LEVELS = 376, 1241, 4


def variants(threads):
    return [("t%d" % t, t, 0) for t in threads] + [("f64", 1, 1)]


def run_pair(orc, k, kw, threads, rows=ROWS, cols=COLS):
    d = synth.make_pair(rows, cols, k)
    out = {}
    for name, nt, red in variants(threads):
        ctx = orc.create(d["K"], d["b"], rows, cols, make_params(orc, **kw), n_frames=2, n_pairs=1)
        ctx.call("set_num_threads", nt)
        ctx.call("set_reduction", red)
        ctx.frame_set_data(0, d["imgA"], d["dispA"])
        ctx.frame_set_template(0)
        ctx.frame_set_data(1, d["imgB"], d["dispB"])
        T, st = ctx.estimate_pose(0, 0, 1)
        out[name] = dict(T=T, its=[s["numIterations"] for s in st], status=[s["status"] for s in st])
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", default="80")
    ap.add_argument("--tolerances", default="timing")
    ap.add_argument("--threads", default="1,2,3,4,5,6,7,8")
    a = ap.parse_args()
    orc = capi.Binding(ge.build_oracle(), "bpvo_orc_")
    kw = dict(descriptor="bitplanes", loss="tukey", levels=LEVELS)
    if a.tolerances == "timing":
        kw.update(parameterTolerance=1e-6, functionTolerance=1e-4, gradientTolerance=1e-6)
    threads = [int(x) for x in a.threads.split(",")]
    for k in [int(x) for x in a.pairs.split(",")]:
        res = run_pair(orc, k, kw, threads)
        base = res["t1"]["T"]
        print(f"pair {k}:")
        for name, r in res.items():
            rot, tr = pose_error(r["T"], base)
            print(f"  {name:4s} its {r['its']} status {[hex(s) for s in r['status']]}  vs t1: {rot:.2e} rad {tr:.2e} m")


if __name__ == "__main__":
    main()
