#!/usr/bin/env python3
"""bench.py — Gauss-Newton iterations/s of the dense photometric alignment hot path on MI355X.

Workload (BASELINE.json configs[3]/[4]): batches of independent KITTI-shaped 1241x376 stereo pairs, BitPlanes
descriptor (8 channels), 4 pyramid levels, Tukey IRLS, AlgorithmParameters() defaults otherwise.  One "step" =
one pass of the hot path over the rank's batch with the inputs already resident in HBM:
    setData(A), setData(B)  (image pyramid + descriptor pyramid)          for every pair
    setTemplate(A)          (saliency, NMS selection, points, Jacobians)  for every pair
    estimatePose(A, B, I)   (coarse-to-fine GN / IRLS to convergence)     for every pair
    + for N > 1: ONE RCCL gather of the 32-float result records to rank 0.
The job is BASELINE.json config 5: ONE batch of --pairs (1024) pairs, seeds 1000 + pair index, split contiguously over the
ranks — strong scaling: `--gpus N` gives every rank 1024 / N pairs (N = 1: the whole batch on one GPU).  `--weak` instead runs
--pairs on EVERY rank.  There is no data-path collective.  value = GN iterations (linearise + solve + pose update, counted
like the reference's _num_fun_evals, bpvo/pose_estimator_gn.h:78) of the whole job per second of step time.

Launch: `python bench.py` (1 GPU), `python bench.py --gpus N` (this process starts the N ranks as children of itself, before it
touches the GPU, and passes rank 0's line through), or the ranks started from outside:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
Every rank checks WORLD_SIZE == --gpus; the line carries `ranks_seen` (an all-reduced count) and the RCCL version.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1024, help="pairs of the job (BASELINE.json config 5: 1024), split over the ranks; with --weak: pairs per rank")
    ap.add_argument("--weak", action="store_true", help="weak scaling: every rank runs --pairs pairs (the job grows with N)")
    ap.add_argument("--pairs-per-gpu", type=int, default=0, help="shorthand for --weak --pairs P")
    ap.add_argument("--rows", type=int, default=376)
    ap.add_argument("--cols", type=int, default=1241)
    ap.add_argument("--descriptor", default="bitplanes", choices=["bitplanes", "intensity"])
    ap.add_argument("--loss", default="tukey", choices=["tukey", "huber", "l2"])
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--interp", default="linear", choices=["linear", "cosine", "cubic", "cubic_hermite"],
                    help="PhotoError interpolation (AlgorithmParameters::interp); the non-linear ones run warp_residual_interp_kernel (profiling runs)")
    ap.add_argument("--strong", action="store_true", help="(the default; kept so that older command lines still parse)")
    ap.add_argument("--tolerances", default="default", choices=["default", "timing"],
                    help="default = AlgorithmParameters() (1e-7 / 1e-6 / 1e-8); timing = the reference's conf/perf_*.cfg (1e-6 / 1e-4 / 1e-6)")
    ap.add_argument("--fixed-iters", type=int, default=0,
                    help="throughput mode: tolerances 0 and maxIterations=K (K+2 linearisations per level); 0 = converge")
    ap.add_argument("--cpu-pairs", type=int, default=40, help="bounded sample for the CPU baseline (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record HIP events at all")
    ap.add_argument("--profile-all", action="store_true", help="HIP events around every kernel (diagnostics, slower)")
    ap.add_argument("--gen-workers", type=int, default=0)
    ap.add_argument("--other-configs", type=int, default=256,
                    help="pairs per batch for the extra lines of BASELINE.json's other single-GPU configs (640x480 intensity/Huber, "
                         "640x480 bit-planes/Tukey) and the B = 1 latency of the headline config; 0 = skip; N = 1 only")
    ap.add_argument("--input-cache", default="", help="directory holding the rendered synthetic inputs of this exact shard; "
                    "written on first use (lets the rocprofv3 runs skip the CPU rendering, which must not fork under the profiler)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL over xGMI; gloo only for functional tests)")
    ap.add_argument("--single-device", action="store_true",
                    help="functional test only: every rank uses GPU 0 (needs --dist-backend gloo)")
    ap.add_argument("--dump-records", default="", help="rank 0 saves the gathered [world * pairs, 32] result records of the last "
                    "step here (.npy); slots [30], [31] of every record carry the rank and the device ordinal that produced it")
    a = ap.parse_args()
    if a.interp != "linear":
        a.over = {"interp": {"cosine": 1, "cubic": 2, "cubic_hermite": 3}[a.interp]}
    return a


from bpvo_amd.bench_configs import make_params, other_configs, stereo_lines  # noqa: E402,F401  (stereo_lines: scripts/stereo_bench.py)


def cpu_baseline(args, batch, n_sample):
    """The CPU oracle (port of the reference path, -O3 -msse4.1 -mavx) on a bounded sample of the same workload."""
    import __graft_entry__ as ge
    from bpvo_amd import capi
    if not os.path.exists(ge.ORACLE_LIB):
        ge.build_oracle()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    p = make_params(orc, args)
    n = min(n_sample, batch["images"].shape[0] // 2)
    out = {}
    # the reference's decomposition offers at most C = 8-way parallelism (channels)
    ncores = min(8, os.cpu_count() or 1)
    for label, threads in (("1", 1), ("all", ncores)):
        ctx = orc.create(batch["K"], batch["b"], args.rows, args.cols, p, n_frames=2 * n, n_pairs=n)
        ctx.call("set_num_threads", threads)
        t0 = time.perf_counter()
        cpu_poses, _ = ctx.batch_run(batch["images"][: 2 * n], batch["disparities"][: 2 * n])
        dt = time.perf_counter() - t0
        out[label] = dict(gn_iters=ctx.total_linearizations(), seconds=dt, threads=threads, poses=cpu_poses)
        ctx.close()
    one = out["1"]
    allc = out["all"]
    return {
        "value": one["gn_iters"] / one["seconds"], "unit": "GN iterations/s", "cores": 1, "kind": "port",
        "sample": f"{n} pairs of the same workload (seeds 1000..{999 + n}), oracle/ C++ restatement single-threaded "
                  f"(= the reference's default build, WITH_TBB OFF): {one['gn_iters']} GN iterations in {one['seconds']:.2f} s",
        "frames_per_s": n / one["seconds"],
        "_poses": one["poses"],
        "all_cores": {"value": allc["gn_iters"] / allc["seconds"], "cores": allc["threads"], "seconds": allc["seconds"], "kind": "port, OpenMP over channels",
                      "host_cores": os.cpu_count(),
                      "note": "OpenMP over the 8 channels / range-split reduction = the reference's TBB decomposition (max 8-way)"},
    }


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks as CHILD processes (python -m torch.distributed.run
    ... bench.py <the same arguments>), before this process has imported torch.cuda or the HIP library — nothing here touches the
    GPU, and nothing is exec'd.  Rank 0's JSON line is the children's stdout, passed through; the exit code is theirs."""
    import socket
    import subprocess
    import torch                       # device_count() does not initialise the GPU on this image
    ndev = torch.cuda.device_count()
    if args.gpus > ndev and not args.single_device:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible (functional test on one device: --single-device --dist-backend gloo)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); pass --gpus {world}")

    from bpvo_amd import synth
    from bpvo_amd.distributed import RECORD_FLOATS, gather_records, records_to_poses, shard_range

    # ---- synthetic inputs for this rank's shard (rendered on the CPU before anything touches the GPU)
    if args.pairs_per_gpu > 0:
        args.weak, args.pairs = True, args.pairs_per_gpu
    if args.weak:
        P = args.pairs
        lo, hi = shard_range(P * world, rank, world)
    else:
        total = args.pairs
        lo, hi = shard_range(total, rank, world)
        P = hi - lo
        if P <= 0 or total % world:
            raise SystemExit("the batch must divide by the number of ranks (strong scaling: --pairs / --gpus pairs per rank)")
    workers = args.gen_workers or max(1, min(16, (os.cpu_count() or 1) // max(1, world)))
    t0 = time.perf_counter()
    cache = os.path.join(args.input_cache, f"synth_{args.rows}x{args.cols}_{lo}_{hi}") if args.input_cache else ""
    if cache and os.path.exists(cache + ".ok"):
        K, b = synth.calibration(args.rows, args.cols)
        batch = dict(K=K, b=b, images=np.load(cache + "_img.npy"), disparities=np.load(cache + "_disp.npy"), T_gt=np.load(cache + "_gt.npy"))
    else:
        batch = synth.make_batch(args.rows, args.cols, hi - lo, first_index=lo, workers=workers)
        if cache:
            os.makedirs(args.input_cache, exist_ok=True)
            np.save(cache + "_img.npy", batch["images"]); np.save(cache + "_disp.npy", batch["disparities"]); np.save(cache + "_gt.npy", batch["T_gt"])
            open(cache + ".ok", "w").close()
    t_gen = time.perf_counter() - t0
    # inputs of the extra configs are rendered now as well: the fork pool must not run after the GPU is initialised
    other_batch = None
    seq640 = None
    if world == 1 and args.other_configs > 0 and (args.rows, args.cols) == (376, 1241):
        other_batch = synth.make_batch(480, 640, args.other_configs, first_index=0, workers=workers)
        seq640 = synth.make_sequence(480, 640, 25, index=21, step_rot=0.004, step_trans=0.03)
        # rectified pairs for the stereo front-end lines (rendered now: no CPU rendering after the GPU is up)
        def stereo_stack(rows_, cols_, n_):
            ps = [synth.make_stereo_pair(rows_, cols_, i) for i in range(n_)]
            return (np.stack([q["left"] for q in ps]), np.stack([q["right"] for q in ps]), ps[0]["K"], ps[0]["b"])
        st_seq = synth.make_stereo_sequence(480, 640, 9, index=23, step_rot=0.004, step_trans=0.03)
        st_k, st_t = stereo_stack(376, 1241, 16), stereo_stack(480, 640, 16)
        seq640["stereo"] = {"batches": {"block matching 1241x376, 128 disparities, batch of 16 pairs": (376, 1241, 128, st_k, "bm"),
                                        "block matching 640x480, 64 disparities, batch of 16 pairs": (480, 640, 64, st_t, "bm"),
                                        "SGM (SgmStereo, conf/kitti_eval.cfg) 1241x376, 128 disparities, batch of 16 pairs": (376, 1241, 128, st_k, "sgm"),
                                        "SGM 640x480, 64 disparities, batch of 16 pairs": (480, 640, 64, st_t, "sgm"),
                                        "SGBM (cv::StereoSGBM as conf/kitti_seq_0.cfg builds it) 1241x376, 128 disparities, batch of 16 pairs": (376, 1241, 128, st_k, "sgbm"),
                                        "SGBM 640x480, 64 disparities, batch of 16 pairs": (480, 640, 64, st_t, "sgbm")},
                            "sequence": (480, 640, 64, st_seq["frames"], st_seq["K"], st_seq["b"])}

    import torch
    import torch.distributed as dist
    import bpvo_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where collective tensors live
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")

    hip = bpvo_amd.load()
    p = make_params(hip, args)
    ctx = hip.create(batch["K"], batch["b"], args.rows, args.cols, p, device=dev_index, n_frames=2 * P, n_pairs=P)

    d_images = torch.from_numpy(batch["images"]).to(dev)
    d_disps = torch.from_numpy(batch["disparities"]).to(dev)
    d_records = torch.zeros((P, RECORD_FLOATS), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()

    gather_s = [0.0, 0]      # seconds this rank spent in the gather (its own clock: includes waiting for the slowest rank), calls

    def step():
        poses, stats = ctx.batch_run_device(P, d_images.data_ptr(), d_disps.data_ptr())
        gathered = None
        if world > 1 or args.dump_records:
            ctx.batch_copy_records_device(d_records.data_ptr(), P)
            d_records[:, 30] = float(rank)         # who produced the record (slots the library leaves at 0)
            d_records[:, 31] = float(dev_index)
            torch.cuda.synchronize()
            tg = time.perf_counter()
            gathered = gather_records(d_records if coll_dev.type == "cuda" else d_records.cpu(), dst=0)
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - tg
            gather_s[1] += 1
        return poses, stats, gathered

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gather_s[0], gather_s[1] = 0.0, 0
    ctx.profiling(0 if args.no_profile else (2 if args.profile_all else 1))   # resets counters; events on the library's streams
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        poses, stats, gathered = step()
    sync_all()
    elapsed = time.perf_counter() - t0

    gn_local = ctx.total_linearizations()
    med_paths = ctx.median_path_counts()
    tap = ctx.tap_cache_counts()
    fused_pts = ctx.fused_point_counts()
    all_kstats = {k["name"]: k for k in ctx.kernel_stats()}
    points_linearized = all_kstats["irls_reduce"]["units"]     # device-side count: sum over linearisations of N
    kstats = all_kstats if not args.no_profile else {}
    # The timed steps run the batch on several estimation lanes (streams; three for bit-planes, two for intensity): faster, but the per-launch durations of one lane then
    # include time shared with the other lane's kernels.  The roofline of warp_residual is therefore taken from ONE more,
    # untimed step of the same workload on a single lane (same kernels, same launches, HIP events on the library's stream);
    # what the events saw inside the timed region is reported next to it (roofline_timed_region).
    kstats_1lane = {}
    fused_1lane = (0, 0)
    if not args.no_profile:
        ctx.set_max_lanes(1)
        ctx.profiling(2 if args.profile_all else 3)     # 3: every warp_residual and irls_reduce launch, so that the averages are rocprofv3's
        ctx.batch_run_device(P, d_images.data_ptr(), d_disps.data_ptr())
        torch.cuda.synchronize()
        kstats_1lane = {k["name"]: k for k in ctx.kernel_stats()}
        fused_1lane = ctx.fused_point_counts()
        ctx.set_max_lanes(0)
    # the reference's own iteration counter (OptimizerStatistics::numIterations, bpvo/pose_estimator_base.h:392-398) summed over
    # pairs and levels of the LAST step; `gn_local` counts linearisations (= _num_fun_evals, pose_estimator_gn.h:78), 1-2 more per level
    numit_last_step = float(stats["numIterations"].sum())
    frame_ms_local = sum(all_kstats[k]["total_ms"] for k in ("pyramid", "descriptor", "saliency_select", "normalization", "template_build"))
    t = torch.tensor([elapsed, float(gn_local), numit_last_step, frame_ms_local, 1.0], dtype=torch.float64, device=coll_dev)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, gn_total, numit_total, ranks_seen = float(tmax[0]), float(tsum[1]), float(tsum[2]), int(round(float(tsum[4])))
    else:
        elapsed_max, gn_total, numit_total, ranks_seen = elapsed, float(gn_local), numit_last_step, 1
    if ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: {ranks_seen} rank(s) took part in the all-reduce, --gpus {args.gpus}")
    # What every rank did, so that the first multi-GPU run can be read: its own step time, the time it spent in the gather (its own clock: a rank
    # that finishes early waits there for the slowest), which Gauss-Newton driver its batch took (team launches / chain), whether a persistent or
    # team launch of its context ever gave up at a barrier and fell back to the chain, and which device it ran on.
    levels_pk, gave_up = ctx.persistent_counts()
    team_launches = ctx.team_counts()
    expect_team = bool(ctx.get_option("team") and ctx.get_option("persistent") and 2 <= P <= ctx.get_option("team_max_pairs") and not args.profile_all)
    my_diag = {"rank": rank, "local_rank": local_rank, "device": dev_index, "device_name": torch.cuda.get_device_name(dev_index),
               "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "ROCR_VISIBLE_DEVICES": os.environ.get("ROCR_VISIBLE_DEVICES"),
               "pairs": P, "ms_per_step": 1e3 * elapsed / args.steps, "gather_ms_per_step": (1e3 * gather_s[0] / gather_s[1]) if gather_s[1] else None,
               "team_launches": int(team_launches), "team_expected": expect_team, "persistent_levels": int(levels_pk), "gave_up": int(gave_up),
               "gn_iterations": int(gn_local)}
    if world > 1:
        diags = [None] * world
        dist.all_gather_object(diags, my_diag)
    else:
        diags = [my_diag]
    silent_fallback = [d["rank"] for d in diags if d["gave_up"] or (d["team_expected"] and d["team_launches"] == 0)]
    if silent_fallback:
        raise SystemExit("bench.py: rank(s) %s fell back from the team / persistent kernel to the four-kernel chain (a device-side barrier timed out, "
                         "or the team kernel never ran where the batch size asks for it) — no line is printed for a run whose ranks did different work: %s"
                         % (silent_fallback, json.dumps(diags)))
    rccl_version = None
    if world > 1 and args.dist_backend == "nccl":
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl_version = "unknown"

    if rank == 0:
        n_pairs_total = P * world
        pose_err = None
        T_gt = batch["T_gt"]
        dT = np.linalg.norm(poses[:, :3, 3].astype(np.float64) - T_gt[:, :3, 3], axis=1)
        pose_err = {"median_trans_err_vs_gt_m": float(np.median(dT)), "max_trans_err_vs_gt_m": float(dT.max())}
        if gathered is not None:
            gp, _, _ = records_to_poses(gathered)
            assert gp.shape[0] == n_pairs_total and np.array_equal(gp[:P, :3, :], poses[:, :3, :])
            if args.dump_records:
                np.save(args.dump_records, gathered.detach().cpu().numpy())

        roofline = None

        def kernel_table(ks):
            out_ = {}
            for name, k in ks.items():
                if k["launches"] == 0:
                    continue
                avg_ms = k["total_ms"] / k["launches"]
                bytes_per_launch = k["units"] * k["bytes_per_unit"] / k["launches"]
                out_[name] = {"launches": int(k["launches"]), "avg_ms": avg_ms, "total_ms": k["total_ms"],
                              "units_per_launch": k["units"] / k["launches"],
                              "algorithmic_GBps": (bytes_per_launch / (avg_ms * 1e-3) / 1e9) if avg_ms > 0 else None}
            return out_
        kernels = kernel_table(kstats)
        kernels_1lane = kernel_table(kstats_1lane)
        roofline_timed = None
        if "warp_residual" in kernels:
            kt = kernels["warp_residual"]
            roofline_timed = {"achieved": kt["algorithmic_GBps"], "frac": kt["algorithmic_GBps"] / HBM_PEAK_GBS, "avg_launch_ms": kt["avg_ms"],
                              "points_per_launch": kt["units_per_launch"],
                              "note": "HIP events inside the timed region: the batch runs on several lanes (three for 8 channels), a launch shares the chip with the other lanes' kernels"}
        if "warp_residual" in kernels_1lane:
            k = kernels_1lane["warp_residual"]
            # HBM bytes per launch from the PMC passes (profiles/collect_profiles.sh): measured per template point on a
            # bounded run of the same kernel, scaled to this run's points per launch
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and args.descriptor == "bitplanes":
                try:
                    traffic = json.load(open(tpath))["warp_residual_hbm_bytes_per_point"] * k["units_per_launch"]
                except Exception:
                    traffic = None
            C_k6 = 8 if args.descriptor == "bitplanes" else 1
            kname, bpp = "warp_residual_kernel<%d>" % C_k6, 18 + 24 * C_k6
            gbps_k6 = k["algorithmic_GBps"]
            if args.interp != "linear":      # the other interpolation types: 4 (kCosine) or 16 taps per point and channel, no tap cache
                kname, bpp = "warp_residual_interp_kernel<%d, %s>" % (C_k6, args.interp), 18 + C_k6 * (4 * (4 if args.interp == "cosine" else 16) + 8)
                gbps_k6 = bpp * k["units_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9
                traffic = None
            roofline = {"bound": "hbm", "kernel": kname,
                        "achieved": gbps_k6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbps_k6 / HBM_PEAK_GBS, "traffic": traffic,
                        "bytes_per_point": bpp,
                        "points_per_launch": k["units_per_launch"], "avg_launch_ms": k["avg_ms"],
                        "measured": "one untimed step of the same workload on a single estimation lane, HIP events around EVERY launch on the "
                                    "library's stream (the timed steps overlap the lanes: roofline_timed_region)"}

        # The second chip-filling kernel, irls_reduce (since round 2 the larger share of the GPU time), in MOVED bytes: what its
        # loads request.  SURVEY.md 8d's algorithmic 2 + 28 C does not describe it — the Jacobian rows are recomputed, not read,
        # and a workspace with a frozen scale takes the fused path (residuals recomputed from the taps):
        #   plain  point 16 + valid 1 + residuals 4 C + gradients 8 C                       = 17 + 12 C  (113 B at C = 8)
        #   fused  point 16 + template pixels 4 C + gradients 8 C + taps 16 C (+ 4 B key)   = 16 + 28 C (+ 4) (244 / 240 B)
        # Next to it the HBM bytes per point the PMC passes counted on this workload (profiles/traffic.json).
        roofline_irls = None
        tjson = {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tjson = json.load(open(tpath))
            except Exception:
                tjson = {}
        if "irls_reduce" in kernels_1lane and args.descriptor == "bitplanes":
            k = kernels_1lane["irls_reduce"]
            C_ = 8
            fused_1, total_1 = fused_1lane
            frac_fused = fused_1 / total_1 if total_1 else 0.0
            moved_per_point = (17 + 12 * C_) * (1.0 - frac_fused) + (16 + 28 * C_ + 4) * frac_fused
            gbps = moved_per_point * k["units_per_launch"] / (k["avg_ms"] * 1e-3) / 1e9
            t_pp = tjson.get("irls_reduce_hbm_bytes_per_point")
            roofline_irls = {"bound": "hbm", "kernel": "irls_reduce_both_kernel<" + args.loss + ">", "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": gbps / HBM_PEAK_GBS, "traffic": (t_pp * k["units_per_launch"]) if t_pp else None,
                             "bytes_per_point": moved_per_point, "bytes_per_point_plain": 17 + 12 * C_, "bytes_per_point_fused": 16 + 28 * C_ + 4,
                             "fused_fraction_of_points": frac_fused, "points_per_launch": k["units_per_launch"], "avg_launch_ms": k["avg_ms"],
                             "launches": k["launches"], "accounting": "moved (requested) bytes, not SURVEY 8d's algorithmic 226 B: Jacobians are recomputed, a third of the points recompute their residuals",
                             "measured": "the same single-lane pass, HIP events around EVERY launch"}

        if roofline is not None and roofline_irls is not None:
            # north_star names warp + residual, so `roofline` is that kernel; the DOMINANT kernel of the step is the reduction, and its figures ride along
            k6t, k8t = kernels_1lane["warp_residual"]["total_ms"], kernels_1lane["irls_reduce"]["total_ms"]
            all_t = sum(v["total_ms"] for v in kernels_1lane.values())
            t_pp = tjson.get("irls_reduce_hbm_bytes_per_point")
            cnt_gbps = (t_pp * roofline_irls["points_per_launch"] / (roofline_irls["avg_launch_ms"] * 1e-3) / 1e9) if t_pp else None
            roofline["dominant"] = {"kernel": "irls_reduce_both_kernel", "avg_launch_ms": roofline_irls["avg_launch_ms"], "launches": roofline_irls["launches"],
                                    "share_of_timed_kernel_ms": k8t / all_t if all_t else None, "warp_residual_share": k6t / all_t if all_t else None,
                                    "frac_counter_bytes": (cnt_gbps / HBM_PEAK_GBS) if cnt_gbps else None, "counter_bytes_per_point": t_pp,
                                    "frac_moved_bytes": roofline_irls["frac"], "moved_bytes_per_point": roofline_irls["bytes_per_point"]}

        # Both chip-filling kernels of the Gauss-Newton loop together, in HBM bytes the counters saw (PMC passes on this workload, bytes per
        # point from profiles/traffic.json) over the sum of their launch durations in the single-lane pass
        gn_kernels = None
        if roofline is not None and roofline_irls is not None and tjson.get("warp_residual_hbm_bytes_per_point") and tjson.get("irls_reduce_hbm_bytes_per_point"):
            k6, k8 = kernels_1lane["warp_residual"], kernels_1lane["irls_reduce"]
            hbm_bytes = tjson["warp_residual_hbm_bytes_per_point"] * k6["units_per_launch"] * k6["launches"] + \
                tjson["irls_reduce_hbm_bytes_per_point"] * k8["units_per_launch"] * k8["launches"]
            secs = (k6["total_ms"] + k8["total_ms"]) * 1e-3
            gn_kernels = {"kernels": ["warp_residual", "irls_reduce"], "hbm_bytes": hbm_bytes, "kernel_seconds": secs, "achieved": hbm_bytes / secs / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_bytes / secs / 1e9 / HBM_PEAK_GBS,
                          "share_of_step": secs / (elapsed / args.steps),
                          "note": "HBM traffic by the PMC counters over the summed launch durations of the two kernels (single-lane pass); "
                                  "share_of_step compares that sum with one timed step (lanes overlapped)"}

        cpu = None
        pose_vs_cpu = None
        if args.cpu_pairs > 0 and world == 1:      # the CPU leg runs at N = 1 only (the other ranks would wait for it)
            cpu = cpu_baseline(args, batch, args.cpu_pairs)
            # pose agreement of the GPU path with the CPU path on the pairs both ran (BASELINE.json: "pose RMSE vs ref")
            cp = cpu.pop("_poses").astype(np.float64)
            gp = poses[: cp.shape[0]].astype(np.float64)
            E = np.einsum("nji,njk->nik", cp[:, :3, :3], gp[:, :3, :3])                    # R_cpu^T R_gpu
            w = 0.5 * np.stack([E[:, 2, 1] - E[:, 1, 2], E[:, 0, 2] - E[:, 2, 0], E[:, 1, 0] - E[:, 0, 1]], axis=1)
            rot = np.arcsin(np.minimum(1.0, np.linalg.norm(w, axis=1)))
            tr = np.linalg.norm(cp[:, :3, 3] - gp[:, :3, 3], axis=1)
            pose_vs_cpu = {"pairs": int(cp.shape[0]), "rot_rmse_rad": float(np.sqrt(np.mean(rot ** 2))), "rot_max_rad": float(rot.max()),
                           "trans_rmse_m": float(np.sqrt(np.mean(tr ** 2))), "trans_max_m": float(tr.max()),
                           "tolerance": "1e-4 rad / 1e-3 m (BASELINE.json north_star)"}

        others = None
        if other_batch is not None:
            # the main context goes first: a process holds a handful of hardware queues, and a second context's two estimation
            # lanes (streams) end up sharing them with the first one's — measured: the 128-pair shard 505 k instead of 666 k GN it/s
            ctx.close()
            del d_images, d_disps, d_records
            torch.cuda.empty_cache()
            args._resident_ms = 1e3 * elapsed_max / args.steps
            others = other_configs(hip, torch, dev, dev_index, args, batch, other_batch, seq640)

        # Strong scaling of config 5 (ONE 1024-pair batch over G GPUs), PROJECTED from this one GPU: rank r of a G-GPU job runs the
        # 1024 / G-pair shard timed above, the ranks do not talk to each other until the single gather of 32-float records (128 KB in all),
        # so the job's rate is G x the shard's rate minus that gather.  No xGMI, no RCCL in these numbers.
        projected = None
        if others is not None and world == 1 and P == 1024:
            shard_rate = {1: gn_total / elapsed_max}
            for k, v in others.items():
                if k.startswith("config-5 shard:"):
                    shard_rate[1024 // v["pairs"]] = v["value"]
            projected = {"note": "one-GPU projection of `--gpus G` (strong scaling: the 1024-pair batch split over G ranks): G x the rate of a "
                                 "1024 / G-pair shard measured on THIS GPU; no xGMI, no RCCL gather (128 KB per job) in it",
                         "points": [{"gpus": g, "pairs_per_rank": 1024 // g, "gn_it_per_s_per_rank": shard_rate[g], "gn_it_per_s_job": g * shard_rate[g],
                                     "efficiency_vs_1_gpu": shard_rate[g] / shard_rate[1]} for g in sorted(shard_rate)]}

        iters = stats["numIterations"].astype(np.float64)
        out = {
            "metric": "GN iterations/s (dense photometric alignment, 1241x376 bit-planes 8ch, 4 levels, Tukey IRLS; one iteration = one "
                      "linearisation + solve + pose update, counted like the reference's _num_fun_evals)"
            if (args.rows, args.cols, args.descriptor) == (376, 1241, "bitplanes") else "GN iterations/s",
            "value": gn_total / elapsed_max, "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "dist_backend": (args.dist_backend if world > 1 else None), "ranks_seen": ranks_seen, "rccl_version": rccl_version,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak" if args.weak else "strong", "vs_baseline": None,
            "dtype": "f32 (+f64 projection/interpolation)", "data": "synthetic",
            "config": {"workload": f"batch of {n_pairs_total} independent {args.cols}x{args.rows} stereo pairs "
                                   f"({P} per GPU), {args.descriptor} descriptor, {args.levels} pyramid levels, {args.loss} loss, "
                                   + (("converge with AlgorithmParameters() tolerances" if args.tolerances == "default" else
                                       "converge with the tolerances of conf/perf_*.cfg (1e-6 / 1e-4 / 1e-6)") if args.fixed_iters == 0 else
                                      f"fixed {args.fixed_iters} iterations/level (tolerances 0)"),
                       "pairs_per_gpu": P, "sharding": f"pairs/{world} ranks, one RCCL gather of 32-float records" if world > 1 else "single GPU",
                       "step": "setData(A,B) + setTemplate(A) + estimatePose(A,B) per pair, inputs resident in HBM"},
            "frames_per_s": 2.0 * n_pairs_total * args.steps / elapsed_max,
            "pairs_per_s": n_pairs_total * args.steps / elapsed_max,
            "gn_iterations_per_step": gn_total / args.steps,
            "numIterations_per_s": numit_total * args.steps / elapsed_max,
            "numIterations_note": "sum over pairs and levels of OptimizerStatistics::numIterations (the reference's reported count: 1-2 below "
                                  "the linearisations of a level) per second — the like-for-like figure next to `value`",
            "roofline_irls_reduce": roofline_irls,
            "gn_kernels_hbm": gn_kernels,
            "points_linearized_rank0": points_linearized,
            "mean_iterations_per_level": [float(x) for x in iters.mean(axis=0)],
            "pose_check": pose_err,
            "pose_vs_cpu": pose_vs_cpu,
            "median_selections": {"bracketed": med_paths[0], "full": med_paths[1]},
            "tap_cache": {"hits": tap[0], "lookups": tap[1], "hit_rate": (tap[0] / tap[1]) if tap[1] else None,
                          "first_8_linearisations_of_a_level": {"hits": tap[2], "lookups": tap[3], "hit_rate": (tap[2] / tap[3]) if tap[3] else None},
                          "later_linearisations": {"hit_rate": ((tap[0] - tap[2]) / (tap[1] - tap[3])) if tap[1] > tap[3] else None},
                          "note": "lookups = valid template points of the (workspace, level) linearisations that use the tap cache: the dense pyramid levels of a batch gather straight from the descriptor"},
            "fused_path": {"points": fused_pts[0], "of": fused_pts[1],
                           "note": "linearisations with a frozen robust scale: residuals recomputed inside irls_reduce, warp_residual skips them"},
            "roofline": roofline,
            "roofline_timed_region": roofline_timed,
            "kernels": kernels_1lane,
            "kernels_timed_region": kernels,
            "cpu_baseline": cpu,
            "other_configs": others,
            "projected_strong_scaling": projected,
            "setup": {"synth_seconds": t_gen, "gen_workers": workers},
            "ranks": diags,
        }
        # ONE JSON line on stdout (the contract), short enough (< 7 KB) that a tail of the output holds all of it: the contract's keys, `roofline`
        # (warp + residual, the kernel north_star names, with the dominant kernel's figures inside), `cpu_baseline`, what every rank did, the
        # config-5 shards, the single pair, the other configurations as one number each.  The full record (kernel tables, notes, every
        # other_configs line) goes to stderr as a second JSON line and to gpurun_out/bench_details.json.
        def rnd(x, nd=4):
            if isinstance(x, float):
                return float(("%." + str(nd) + "g") % x)
            if isinstance(x, dict):
                return {k: rnd(v, nd) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [rnd(v, nd) for v in x]
            return x
        head = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")}
        head["metric"] = "GN iterations/s (1241x376 bit-planes 8ch, 4 levels, Tukey; linearise + solve + update = 1, the reference's _num_fun_evals)" \
            if (args.rows, args.cols, args.descriptor) == (376, 1241, "bitplanes") else "GN iterations/s"
        for k in ("frames_per_s", "gn_iterations_per_step", "ranks_seen", "dist_backend", "rccl_version", "points_linearized_rank0"):
            head[k] = out[k]
        head["fused_path"] = {k: out["fused_path"][k] for k in ("points", "of")}
        if roofline_timed is not None:
            head["roofline_timed_region"] = {k: roofline_timed[k] for k in ("achieved", "frac", "avg_launch_ms")}
        head["kernels"] = {n_: {"launches": v["launches"], "avg_ms": v["avg_ms"], "algorithmic_GBps": v["algorithmic_GBps"]} for n_, v in kernels_1lane.items()}
        if roofline is not None:
            rl = dict(roofline)
            rl["measured"] = "HIP events around every launch, one untimed single-lane step"
            head["roofline"] = rl
        else:
            head["roofline"] = None
        if cpu is not None:
            head["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                    "sample": f"{min(args.cpu_pairs, P)} pairs of the same workload, oracle/ restatement, 1 thread",
                                    "all_cores": {"value": cpu["all_cores"]["value"], "cores": cpu["all_cores"]["cores"]}}
        else:
            head["cpu_baseline"] = None
        if pose_vs_cpu is not None:
            head["pose_vs_cpu"] = {k: pose_vs_cpu[k] for k in ("pairs", "rot_max_rad", "trans_max_m")}
        ms = [d["ms_per_step"] for d in diags]
        head["ranks"] = {"ms_per_step_min": min(ms), "ms_per_step_max": max(ms), "slowest_rank": int(np.argmax(ms)), "dist_backend": out["dist_backend"],
                         "ranks_seen": ranks_seen, "rccl_version": rccl_version,
                         "per_rank": [{k: d[k] for k in ("rank", "device", "HIP_VISIBLE_DEVICES", "pairs", "ms_per_step", "gather_ms_per_step", "team_launches", "gave_up")} for d in diags]}
        if others is not None:
            shards, brief = {}, {}
            for k, v in others.items():
                if k.startswith("config-5 shard:"):
                    shards[str(v["pairs"])] = {"value": v["value"], "of_headline": v["value"] / out["value"], "ms_per_step": v["ms_per_step"]}
                elif "one pair per call" in k:
                    head["single_pair"] = {"value": v["value"], "ms_per_pair": v["ms_per_step"], "us_per_linearisation": v["us_per_linearisation_per_pair"]}
                elif k == "stereo front-end":
                    brief["stereo ms/frame"] = {kk.split(",")[0]: vv.get("ms_per_frame") for kk, vv in v.items()}
                elif "addFrame" in k:
                    brief[k.replace("parameters of ", "")] = {"ms_per_frame": v["ms_per_frame"]}
                elif "HOST buffers" in k:
                    brief["1024 pairs from host buffers"] = {"value": v["value"], "vs_resident_inputs": v["vs_resident_inputs"], "upload_GBps": v["upload"]["GBps"]}
                else:
                    e = {"value": v["value"], "pairs": v["pairs"], "us_per_linearisation_per_pair": v["us_per_linearisation_per_pair"]}
                    if v.get("residual_kernel"):
                        rk = v["residual_kernel"]
                        e["residual_kernel"] = {"kernel": rk["kernel"], "avg_launch_ms": rk["avg_launch_ms"], "bytes_per_point": rk["bytes_per_point"],
                                                "achieved_GBps": rk["achieved_GBps"], "frac_of_hbm_peak": rk["frac_of_hbm_peak"]}
                    brief[k.replace("the parameters of ", "")] = e
            head["config5_shards"] = shards
            head["other_configs"] = brief
        head["details"] = "second JSON line on stderr; gpurun_out/bench_details.json"
        line = json.dumps(rnd(head))
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_details.json"), "w") as f:
                f.write(json.dumps(out) + "\n")
        except OSError:
            pass
        print(json.dumps({"details": out}), file=sys.stderr, flush=True)
        print(line, flush=True)

    if ctx.h:
        ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
