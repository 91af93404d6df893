"""The extra lines of bench.py's record (`other_configs`): BASELINE.json's other single-GPU configurations as batches, the config-5 shards a rank
of a 2 / 4 / 8-GPU job runs, one pair per call, host-buffer batches, the reference's timing tolerances, sequential addFrame with the
reference's two timing configurations, and the stereo front-end.  Same step definition as the headline (bench.py); product library only."""
from __future__ import annotations

import time

import numpy as np


def make_params(binding, args):
    from bpvo_amd import capi
    p = binding.default_params()
    p.numPyramidLevels = args.levels
    p.descriptor = capi.DESC_BITPLANES if args.descriptor == "bitplanes" else capi.DESC_INTENSITY
    p.lossFunction = {"tukey": capi.LOSS_TUKEY, "huber": capi.LOSS_HUBER, "l2": capi.LOSS_L2}[args.loss]
    p.verbosity = capi.VERB_SILENT
    if getattr(args, "tolerances", "default") == "timing":
        p.parameterTolerance = 1e-6
        p.functionTolerance = 1e-4
        p.gradientTolerance = 1e-6
    if args.fixed_iters > 0:
        p.maxIterations = args.fixed_iters
        p.parameterTolerance = 0.0
        p.functionTolerance = 0.0
        p.gradientTolerance = 0.0
    for k, v in (getattr(args, "over", None) or {}).items():      # further AlgorithmParameters fields (the conf/*.cfg lines)
        setattr(p, k, v)
    return p



HBM_PEAK_GBS = 8000.0


def timed_batch(hip, torch, dev, dev_index, batch, rows, cols, n, descriptor, levels, loss, steps=3, warmup=1, tolerances="default", want_stages=False,
                over=None, residual_kernel_taps=0):
    """GN iterations/s of one more configuration (same step definition as the headline, inputs resident in HBM).
    over: more AlgorithmParameters fields.  residual_kernel_taps (4 = kCosine, 16 = kCubic / kCubicHermite): one more, untimed step on a single
    lane with HIP events around every warp + residual launch -> the kernel's average launch and its algorithmic bytes / s (18 + C (4 taps + 8)
    B per point: point 16, valid 1 (+1), per channel the taps, the template pixel and the residual) against the HBM peak."""
    from types import SimpleNamespace
    p = make_params(hip, SimpleNamespace(levels=levels, descriptor=descriptor, loss=loss, fixed_iters=0, tolerances=tolerances, over=over))
    ctx = hip.create(batch["K"], batch["b"], rows, cols, p, device=dev_index, n_frames=2 * n, n_pairs=n)
    d_i = torch.from_numpy(batch["images"][: 2 * n]).to(dev)
    d_d = torch.from_numpy(batch["disparities"][: 2 * n]).to(dev)
    for _ in range(warmup):
        ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
    ctx.profiling(1 if want_stages else 0)   # events around the frame stages + 1 in 5 warp_residual launches (2-3 % of the step)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        poses, stats = ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gn = ctx.total_linearizations()
    frame_ms = None
    if want_stages:
        ks = {k["name"]: k for k in ctx.kernel_stats()}
        frame_ms = sum(ks[k]["total_ms"] for k in ("pyramid", "descriptor", "saliency_select", "normalization", "template_build")) / steps
    dT = np.linalg.norm(poses[:, :3, 3].astype(np.float64) - batch["T_gt"][:n, :3, 3], axis=1)
    kernel = None
    if residual_kernel_taps:
        C = 8 if descriptor == "bitplanes" else 1
        ctx.set_max_lanes(1)
        ctx.profiling(3)
        ctx.batch_run_device(n, d_i.data_ptr(), d_d.data_ptr())
        torch.cuda.synchronize()
        k = {q["name"]: q for q in ctx.kernel_stats()}["warp_residual"]
        bpp = 18 + C * (4 * residual_kernel_taps + 8)
        if k["launches"]:
            avg_ms = k["total_ms"] / k["launches"]
            gbps = bpp * (k["units"] / k["launches"]) / (avg_ms * 1e-3) / 1e9
            kernel = {"kernel": "warp_residual_interp_kernel<%d>" % C, "launches": int(k["launches"]), "avg_launch_ms": avg_ms, "points_per_launch": k["units"] / k["launches"],
                      "bytes_per_point": bpp, "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBS,
                      "note": "algorithmic bytes (every tap counted once per point; neighbouring points share most of their 4 x 4 footprints in L2) over "
                              "HIP events around every launch of one untimed single-lane step"}
    ctx.close()
    return {"pairs": n, "value": gn / dt, "unit": "GN iterations/s", "frames_per_s": 2.0 * n * steps / dt, "ms_per_step": 1e3 * dt / steps,
            "us_per_linearisation_per_pair": 1e6 * dt * n / max(1, gn),      # (a pair's step time over its linearisations: the B = 1 latency figure)
            "frame_stage_ms_per_step": frame_ms,
            "gn_iterations_per_pair": gn / (steps * n), "numIterations_per_pair": float(stats["numIterations"].sum()) / n,
            "median_trans_err_vs_gt_m": float(np.median(dT)), "residual_kernel": kernel}


def timed_batch_host(hip, torch, dev_index, batch, rows, cols, n, descriptor, levels, loss, resident_ms, steps=3, warmup=1):
    """The same step with the inputs in HOST memory (what a drop-in caller hands over): bpvo_hip_batch_run(..., on_device = 0).  The library
    stages chunks of 16 pairs in pinned memory with worker threads and uploads them on streams of their own while the lanes work on the
    chunks that have landed; the current frames' disparities (40 % of the bytes) never cross the bus.  NEVER part of `value`."""
    from types import SimpleNamespace
    p = make_params(hip, SimpleNamespace(levels=levels, descriptor=descriptor, loss=loss, fixed_iters=0, tolerances="default"))
    ctx = hip.create(batch["K"], batch["b"], rows, cols, p, device=dev_index, n_frames=2 * n, n_pairs=n)
    imgs, disps = batch["images"][: 2 * n], batch["disparities"][: 2 * n]
    for _ in range(warmup):
        ctx.batch_run(imgs, disps)
    ctx.profiling(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ups = []
    for _ in range(steps):
        ctx.batch_run(imgs, disps)
        ups.append(ctx.upload_stats())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gn = ctx.total_linearizations()
    ctx.close()
    up_s = float(np.mean([u[0] for u in ups])); up_b = ups[-1][1]
    return {"pairs": n, "value": gn / dt, "unit": "GN iterations/s", "ms_per_step": 1e3 * dt / steps,
            "vs_resident_inputs": (1e3 * dt / steps) / resident_ms if resident_ms else None,
            "upload": {"bytes_per_step": up_b, "seconds": up_s, "GBps": up_b / up_s / 1e9 if up_s > 0 else None,
                       "note": "6 worker threads copy chunks of 16 pairs into pinned slots, H2D copies on one copy stream, wall time until the last chunk has landed; it runs under the compute of the chunks before it (three groups: 19 %, 50 %, 31 % of the pairs)"},
            "note": "pageable numpy buffers handed to bpvo_hip_batch_run; PCIe-inclusive — reported beside `value`, never as `value`"}


def add_frame_latency(hip, dev_index, seq, which):
    """Sequential VisualOdometry::addFrame on a 640x480 sequence with the parameters of the reference's own timing runs
    (conf/perf_intensity.cfg / conf/perf_bitplanes.cfg as AlgorithmParameters(filename) builds them, bpvo/types.cc:68-107):
    the quantity behind BASELINE.md section 1 (ms per addFrame; there: Tsukuba frames on the author's CPU)."""
    from bpvo_amd import capi
    p = hip.default_params()
    p.numPyramidLevels = 3; p.parameterTolerance = 1e-6; p.functionTolerance = 1e-4; p.gradientTolerance = 1e-6
    p.maxIterations = 50; p.relaxTolerancesForCoarseLevels = 1; p.gradientEstimation = capi.GRAD_CD5
    p.minValidDisparity = 1.0; p.goodPointThreshold = 0.75; p.verbosity = capi.VERB_SILENT
    if which == "tsukuba":
        # conf/tsukuba.cfg as AlgorithmParameters(filename) reads it (`Descriptor = ...` is not the key that is read: Intensity): 3 levels,
        # CubicHermite, 55 iterations, Huber, minSaliency 0.001, NMS radius 0 -> dense templates (~ 300 k points at 640x480), key frames every ~ 3 frames
        p.maxIterations = 55; p.relaxTolerancesForCoarseLevels = 0; p.interp = capi.INTERP_CUBIC_HERMITE
        p.descriptor = capi.DESC_INTENSITY; p.lossFunction = capi.LOSS_HUBER; p.minSaliency = 0.001; p.nonMaxSuppRadius = 0
        p.sigmaPriorToCensusTransform = 0.75; p.sigmaBitPlanes = 1.75
        p.minTranslationMagToKeyFrame = 0.05; p.minRotationMagToKeyFrame = 2.5; p.maxFractionOfGoodPointsToKeyFrame = 0.5
    elif which == "perf_bitplanes":
        p.descriptor = capi.DESC_BITPLANES; p.lossFunction = capi.LOSS_L2
        p.minTranslationMagToKeyFrame = 0.1; p.minRotationMagToKeyFrame = 5.0
        p.sigmaPriorToCensusTransform = 0.75; p.sigmaBitPlanes = 1.6
    else:
        p.descriptor = capi.DESC_INTENSITY; p.lossFunction = capi.LOSS_HUBER; p.minSaliency = 2.5; p.nonMaxSuppRadius = 2
        p.minTranslationMagToKeyFrame = 1000.0; p.minRotationMagToKeyFrame = 1000.0; p.maxFractionOfGoodPointsToKeyFrame = 0.75
    ctx = hip.create(seq["K"], seq["b"], 480, 640, p, device=dev_index, n_frames=3, n_pairs=1)
    frames = seq["frames"]
    ctx.add_frame(*frames[0])
    t0 = time.perf_counter()
    n_key = 0
    for img, disp in frames[1:]:
        n_key += int(ctx.add_frame(img, disp)["isKeyFrame"])
    dt = time.perf_counter() - t0
    ctx.close()
    return {"frames": len(frames) - 1, "ms_per_frame": 1e3 * dt / (len(frames) - 1), "frames_per_s": (len(frames) - 1) / dt,
            "keyframes": n_key, "note": "host image + disparity buffers per call (PCIe included), one frame at a time"}


def stereo_lines(hip, torch, dev, dev_index, stereo_in):
    """Stereo front-end: the block matcher of the reference's default StereoAlgorithm on batches of rectified pairs
    resident in HBM (disparities written to HBM), and sequential addFrame(left, right) from host buffers.  The matcher is integer
    SAD over a 15 x 15 window for every (pixel, disparity): VALU-bound (wave prefix sums + compares), not HBM-bound — two u8 images
    in, one f32 map out."""
    out = {}
    from bpvo_amd import capi
    for name, (rows, cols, ndisp, pairs, algorithm) in stereo_in["batches"].items():
        left, right, K, b = pairs
        n = left.shape[0]
        p = hip.default_params(); p.numPyramidLevels = 1
        ctx = hip.create(K, b, rows, cols, p, device=dev_index, n_frames=1, n_pairs=1)
        sp = ctx.default_stereo_params(ndisp)
        sp.algorithm = capi.STEREO_SGM if algorithm == "sgm" else capi.STEREO_BM
        if algorithm == "sgbm":      # conf/kitti_seq_0.cfg: SemiGlobalBlockMatching, SADWindowSize 7, every other key at the default of its cf.get
            sp = ctx.sgbm_params_from_config(0, ndisp, SADWindowSize=7)
        dl, dr = torch.from_numpy(left).to(dev), torch.from_numpy(right).to(dev)
        dd = torch.empty((n, rows, cols), dtype=torch.float32, device=dev)
        for _ in range(2):
            ctx.stereo_bm_device(n, dl.data_ptr(), dr.data_ptr(), sp, dd.data_ptr())
        torch.cuda.synchronize()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.stereo_bm_device(n, dl.data_ptr(), dr.data_ptr(), sp, dd.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if algorithm == "sgbm":
            valid = float((dd[0] >= 0).float().mean().item())
            width1 = cols - ndisp
            vol = float(n) * rows * width1 * ndisp
            # per (cost pixel, disparity): pixel cost 1 B written, read ~7 x by the row sums (L1 / L2) -> 2 B row sums written and read -> 2 B cost
            # written, read by 5 path families that write 2 B each, read again by the selection: 1 + 2 + 2 + 2 + 2 + 5 * (2 + 2) + 5 * 2 = 39 B
            out[name] = {"frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "pixel_disparities_per_s": vol / dt,
                         "valid_fraction_frame0": valid, "disparities": ndisp, "SADWindowSize": 7, "accounted_GBps": 39 * vol / dt / 1e9,
                         "bound": "per-kernel HBM traffic and stall fractions: profiles/r05_stereo_pmc.txt (rocprofv3 --pmc over scripts/stereo_bench.py)"}
        elif algorithm == "sgm":
            valid = float((dd[0] > 0).float().mean().item())
            # per frame: pixel cost (1 B) written + read 25 x by the box sum (L1 / L2), two 2-byte cost volumes written and read by 4 path
            # passes each together with the 2-byte sum volume (read + write): ~ (1 + 2 + 2 + 2 * 4 * (2 + 2 + 2) + 2 * 2) B per (pixel, disparity)
            vol = float(n) * rows * cols * ndisp
            out[name] = {"frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "pixel_disparities_per_s": vol / dt,
                         "valid_fraction_frame0": valid, "disparities": ndisp,
                         "accounted_GBps": 57 * vol / dt / 1e9,      # ~57 B per (pixel, disparity) by the accounting above, over the whole matcher
                         "bound": "measured, profiles/r04_stereo_pmc.txt (rocprofv3 --pmc over scripts/stereo_bench.py): the scanline kernel "
                                  "(37 % of the matcher) moves 4.2 TB/s of HBM = 0.53 of the peak with its waves issue-stalled 44 % of their cycles "
                                  "(one dependent DPP / packed-min chain per step); the vertical box sums run at 0.91 of the peak; winner-take-all "
                                  "reads its four path volumes at 2.9 TB/s (waves waiting 78 %), the right-view cost moves exactly its volume through "
                                  "LDS tiles at 2.7 TB/s: the matcher as a whole is latency-, not bandwidth-bound"}
        else:
            valid = float((dd[0] >= 0).float().mean().item())
            sums = float(n) * rows * (cols - ndisp + 1) * ndisp
            out[name] = {"frames": n, "frames_per_s": n / dt, "ms_per_frame": 1e3 * dt / n, "window_sums_per_s": sums / dt,
                         "valid_fraction_frame0": valid, "disparities": ndisp, "SADWindowSize": sp.SADWindowSize,
                         "bound": "measured, profiles/r04_stereo_pmc.txt: " + "%.1f" % ((2 + 4) * rows * cols * n / dt / 1e9) + " GB/s of HBM traffic by the accounting (2 u8 "
                                  "images in, 1 f32 map out; counters: 17 - 41 GB/s), VALU active 25 % of the wave cycles, waves waiting 57 % (LDS prefix "
                                  "sums of the integer SAD): an instruction- and LDS-latency-bound integer kernel, three orders below the HBM roofline"}
        ctx.close()
    seq = stereo_in.get("sequence")
    if seq is not None:
        rows, cols, ndisp, frames, K, b = seq
        p = hip.default_params(); p.numPyramidLevels = 3; p.verbosity = 0x22
        p.parameterTolerance, p.functionTolerance, p.gradientTolerance = 1e-6, 1e-4, 1e-6
        ctx = hip.create(K, b, rows, cols, p, device=dev_index, n_frames=3, n_pairs=1)
        sp = ctx.default_stereo_params(ndisp)
        ctx.add_frame_stereo(frames[0][0], frames[0][1], sp)
        t0 = time.perf_counter()
        for l, r in frames[1:]:
            ctx.add_frame_stereo(l, r, sp)
        dt = time.perf_counter() - t0
        out[f"addFrame(left, right) {cols}x{rows}, {ndisp} disparities, intensity / 3 levels"] = {
            "frames": len(frames) - 1, "ms_per_frame": 1e3 * dt / (len(frames) - 1),
            "note": "two u8 host images per call; block matching, setData, estimatePose on the device, the f32 disparity never crosses the bus"}
        ctx.close()
    return out


def other_configs(hip, torch, dev, dev_index, args, batch, other_batch, seq640=None):
    """BASELINE.json configs[1], [2] (640x480) as batches, and configs[3] (one KITTI-shaped pair at a time: the latency-bound
    B = 1 case the roofline section of SURVEY.md asks to report next to the batched one)."""
    n = args.other_configs
    out = {
        "640x480 intensity, 4 levels, huber": timed_batch(hip, torch, dev, dev_index, other_batch, 480, 640, n, "intensity", 4, "huber", steps=10, warmup=2),
        "640x480 bitplanes, 4 levels, tukey": timed_batch(hip, torch, dev, dev_index, other_batch, 480, 640, n, "bitplanes", 4, "tukey", steps=6, warmup=2),
        "1241x376 bitplanes, 4 levels, tukey, one pair per call (B = 1)":
            timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, 1, args.descriptor, args.levels, args.loss, steps=40, warmup=5),   # (3 ms steps: 40 of them)
    }
    # The other interpolation types of PhotoError (bpvo/photo_error.cc:391-444; warp_residual_interp_kernel): conf/tsukuba.cfg:45 selects
    # CubicHermite — its parameter set (Intensity as the file is read, 3 levels, 55 iterations, 1e-6 / 1e-4 / 1e-6, Huber, CD5, minSaliency
    # 0.001, NMS radius 0 -> off, sigma_ct 0.75, sigma_bp 1.75; types.cc:68-107 for the file constructor's other defaults) — and the
    # headline shape (bit-planes, Tukey, 4 levels) with kCubic
    from bpvo_amd import capi as _capi
    tsukuba = dict(interp=_capi.INTERP_CUBIC_HERMITE, maxIterations=55, parameterTolerance=1e-6, functionTolerance=1e-4, gradientTolerance=1e-6,
                   gradientEstimation=_capi.GRAD_CD5, minSaliency=0.001, nonMaxSuppRadius=0, minNumPixelsForNonMaximaSuppression=76800,
                   relaxTolerancesForCoarseLevels=0, sigmaPriorToCensusTransform=0.75, sigmaBitPlanes=1.75, minValidDisparity=1.0, goodPointThreshold=0.75,
                   minTranslationMagToKeyFrame=0.05, minRotationMagToKeyFrame=2.5, maxFractionOfGoodPointsToKeyFrame=0.5)
    out["640x480 intensity, 3 levels, huber, CubicHermite: the parameters of conf/tsukuba.cfg"] = \
        timed_batch(hip, torch, dev, dev_index, other_batch, 480, 640, n, "intensity", 3, "huber", steps=6, warmup=2, over=tsukuba, residual_kernel_taps=16)
    out["1241x376 bitplanes, 4 levels, tukey, kCubic interpolation"] = \
        timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, min(n, batch["images"].shape[0] // 2), args.descriptor, args.levels, args.loss, steps=3, warmup=1,
                    over=dict(interp=_capi.INTERP_CUBIC), residual_kernel_taps=16)
    npairs = batch["images"].shape[0] // 2
    if npairs >= 128:
        # BASELINE config 5 as it is really sharded: 1024 pairs over 8 GPUs = 128 pairs per GPU (strong scaling; `--gpus 8` runs
        # exactly this on every rank)
        out["config-5 shard: 128 of the 1024 pairs (1241x376 bitplanes, 4 levels, tukey) on one GPU"] = \
            timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, 128, args.descriptor, args.levels, args.loss, steps=10, warmup=2)
    if npairs >= 1024:
        # ... and what a rank of the 4- and 2-GPU jobs holds, so that the whole strong-scaling curve of config 5 is on this record
        out["config-5 shard: 256 of the 1024 pairs on one GPU (a rank of the 4-GPU job)"] = \
            timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, 256, args.descriptor, args.levels, args.loss, steps=6, warmup=2)
        out["config-5 shard: 512 of the 1024 pairs on one GPU (a rank of the 2-GPU job)"] = \
            timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, 512, args.descriptor, args.levels, args.loss, steps=4, warmup=1)
    if npairs >= 1024 and (args.rows, args.cols, args.descriptor, args.levels, args.loss) == (376, 1241, "bitplanes", 4, "tukey"):
        out["1024 pairs handed over in HOST buffers (upload pipeline)"] = \
            timed_batch_host(hip, torch, dev_index, batch, args.rows, args.cols, 1024, args.descriptor, args.levels, args.loss, args._resident_ms)
    if npairs >= 1024:
        # the reference's own timing tolerances (conf/perf_*.cfg: 1e-6 / 1e-4 / 1e-6, 3 levels): ~7x fewer iterations per level than the
        # AlgorithmParameters() defaults, so the per-frame stages are about half of the step
        out["1241x376 bitplanes, 3 levels, tukey, tolerances of conf/perf_bitplanes.cfg (1e-6/1e-4/1e-6), 1024 pairs"] = \
            timed_batch(hip, torch, dev, dev_index, batch, args.rows, args.cols, 1024, args.descriptor, 3, args.loss, steps=3, warmup=1, tolerances="timing", want_stages=True)
    if seq640 is not None and seq640.get("stereo") is not None:
        out["stereo front-end"] = stereo_lines(hip, torch, dev, dev_index, seq640["stereo"])
    if seq640 is not None:
        out["addFrame 640x480, parameters of conf/perf_intensity.cfg"] = add_frame_latency(hip, dev_index, seq640, "perf_intensity")
        out["addFrame 640x480, parameters of conf/perf_bitplanes.cfg"] = add_frame_latency(hip, dev_index, seq640, "perf_bitplanes")
        out["addFrame 640x480, parameters of conf/tsukuba.cfg"] = add_frame_latency(hip, dev_index, seq640, "tsukuba")
    return out


