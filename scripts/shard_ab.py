#!/usr/bin/env python3
"""A/B of run-time settings on one batch size, every variant in a process of its own (several settings are read once per process).

  python scripts/shard_ab.py --pairs 128 --steps 10 -- "" "lanes=1,stagger=0" "team_max_pairs=128" "BPVO_AB_LIB=bpvo_amd/csrc/exp/libbpvo_hip_x.so,lanes=3"

A variant is a comma-separated list of settings: lower-case keys are library options (bpvo_hip_set_option, handed over through
BPVO_HIP_OPTIONS), upper-case keys are environment variables of the child process (BPVO_AB_LIB, GPU_MAX_HW_QUEUES, ...).

The parent renders the synthetic inputs once (no GPU in the parent: children are plain child processes), every child loads them, runs
`warmup` + `steps` steps of bpvo_hip_batch_run on device-resident inputs and prints GN iterations/s — bench.py's step, nothing else
around it.  `--ref-pairs N` also measures an N-pair batch (the 1024-pair headline) with every variant, for the shard / headline ratio.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np
    import torch
    import bpvo_amd
    from bpvo_amd import capi
    d = np.load(a.child)
    # BPVO_AB_LIB: an experimental build of the library (scripts/build_exp.sh) instead of the product's
    hip = capi.Binding(os.path.join(ROOT, os.environ["BPVO_AB_LIB"]), "bpvo_hip_") if os.environ.get("BPVO_AB_LIB") else bpvo_amd.load()
    p = hip.default_params()
    p.numPyramidLevels = a.levels
    p.descriptor = capi.DESC_BITPLANES if a.descriptor == "bitplanes" else capi.DESC_INTENSITY
    p.lossFunction = {"tukey": capi.LOSS_TUKEY, "huber": capi.LOSS_HUBER, "l2": capi.LOSS_L2}[a.loss]
    p.verbosity = capi.VERB_SILENT
    if a.tolerances == "timing":
        p.parameterTolerance, p.functionTolerance, p.gradientTolerance = 1e-6, 1e-4, 1e-6
    n = a.pairs
    rows, cols = d["images"].shape[1:]
    ctx = hip.create(d["K"], float(d["b"]), rows, cols, p, device=0, n_frames=2 * n, n_pairs=n)
    dev = torch.device("cuda", 0)
    di = torch.from_numpy(d["images"][: 2 * n]).to(dev)
    dd = torch.from_numpy(d["disparities"][: 2 * n]).to(dev)
    hi, hd = d["images"][: 2 * n], d["disparities"][: 2 * n]
    run = (lambda: ctx.batch_run(hi, hd)) if a.host else (lambda: ctx.batch_run_device(n, di.data_ptr(), dd.data_ptr()))
    for _ in range(a.warmup):
        run()
    ctx.profiling(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        poses, stats = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gn = ctx.total_linearizations()
    bracketed, full = ctx.median_path_counts()      # exact-median selections from the bracket's candidates / over all keys (level starts + bracket misses)
    print(json.dumps(dict(pairs=n, value=gn / dt, ms_per_step=1e3 * dt / a.steps, gn_per_step=gn / a.steps, median_bracketed=bracketed, median_full=full,
                          checksum=float(np.abs(poses.astype(np.float64)).sum()))))
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--ref-pairs", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=376)
    ap.add_argument("--cols", type=int, default=1241)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--descriptor", default="bitplanes")
    ap.add_argument("--loss", default="tukey")
    ap.add_argument("--tolerances", default="default")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--child", default="")
    ap.add_argument("--host", action="store_true", help="hand the library HOST buffers (the upload pipeline) instead of device-resident inputs")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a)
    import numpy as np
    from bpvo_amd import synth
    nmax = max(a.pairs, a.ref_pairs)
    path = f"/tmp/shard_ab_{a.rows}x{a.cols}_{nmax}.npz"
    if not os.path.exists(path):
        b = synth.make_batch(a.rows, a.cols, nmax, first_index=0, workers=min(16, os.cpu_count() or 1))
        np.savez(path, images=b["images"], disparities=b["disparities"], K=b["K"], b=b["b"])
    variants = a.variants or [""]
    sizes = [a.pairs] + ([a.ref_pairs] if a.ref_pairs else [])
    for rep in range(a.repeat):
        for v in variants:
            env = dict(os.environ)
            opts = []
            for kv in filter(None, v.split(",")):
                k, val = kv.split("=", 1)
                if k.islower():
                    opts.append(kv)
                else:
                    env[k] = val
            if opts:
                env["BPVO_HIP_OPTIONS"] = ",".join(opts)
            out = []
            for n in sizes:
                steps = a.steps if n == a.pairs else max(2, a.steps // 4)
                cmd = [sys.executable, os.path.abspath(__file__), "--child", path, "--pairs", str(n), "--steps", str(steps), "--warmup", str(a.warmup),
                       "--levels", str(a.levels), "--descriptor", a.descriptor, "--loss", a.loss, "--tolerances", a.tolerances] + (["--host"] if a.host else [])
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                out.append(json.loads(line[-1]) if line else dict(error=(r.stdout + r.stderr)[-400:]))
            msg = " | ".join(f"{o.get('pairs')}: {o.get('value', 0) / 1e3:8.1f} k GN it/s {o.get('ms_per_step', 0):7.2f} ms, full selections {o.get('median_full', 0)} of {o.get('median_full', 0) + o.get('median_bracketed', 0)}, checksum {o.get('checksum', 0):.9g}" if "value" in o else str(o) for o in out)
            if len(out) == 2 and "value" in out[0] and "value" in out[1]:
                msg += f" | ratio {out[0]['value'] / out[1]['value']:.3f}"
            print(f"[{rep}] {v or 'default':60s} {msg}", flush=True)


if __name__ == "__main__":
    main()
