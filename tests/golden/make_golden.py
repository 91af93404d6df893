#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run in the build container: `python tests/golden/make_golden.py`).

The reference itself cannot be built or imported here (Eigen / OpenCV / TBB absent, no Python in the
reference) and its tests hold no golden values, so these vectors are the ORACLE's outputs on committed inputs.  They
pin the oracle against accidental change (CPU suite) and are a second, file-based parity target for the HIP path (GPU
suite); they do not pin the oracle against the reference — DESIGN.md says "parity unpinned" for that reason.

Each case stores the inputs (u8 images, f32 disparities, K, baseline, parameters) and per-stage outputs as data.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402
from bpvo_amd import capi, synth  # noqa: E402
from util import make_params  # noqa: E402

CASES = [
    dict(name="bp_tukey_96x128", rows=96, cols=128, levels=3, descriptor="bitplanes", loss="tukey", index=0),
    dict(name="int_huber_96x128", rows=96, cols=128, levels=3, descriptor="intensity", loss="huber", index=1),
    dict(name="int_l2_1level_96x128", rows=96, cols=128, levels=1, descriptor="intensity", loss="l2", index=2),
    # DisparitySpaceWarp as the warp (set_warp_formulation(2)): disparity-space points, its Jacobian, f32 projection
    dict(name="bp_huber_dspace_96x128", rows=96, cols=128, levels=2, descriptor="bitplanes", loss="huber", index=3, formulation=2),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build_case(orc, c):
    d = synth.make_pair(c["rows"], c["cols"], c["index"])
    p = make_params(orc, descriptor=c["descriptor"], loss=c["loss"], levels=c["levels"])
    ctx = orc.create(d["K"], d["b"], c["rows"], c["cols"], p, n_frames=2, n_pairs=1)
    if c.get("formulation", 0):
        ctx.set_warp_formulation(c["formulation"])
    ctx.frame_set_data(0, d["imgA"], d["dispA"])
    ctx.frame_set_template(0)
    ctx.frame_set_data(1, d["imgB"], d["dispB"])
    out = dict(imgA=d["imgA"], dispA=d["dispA"], imgB=d["imgB"], K=d["K"], baseline=np.float32(d["b"]),
               levels=np.int32(c["levels"]), descriptor=np.bytes_(c["descriptor"]), loss=np.bytes_(c["loss"]))
    if c.get("formulation", 0):
        out["formulation"] = np.int32(c["formulation"])
    T_lin = synth.twist_to_matrix([0.003, -0.002, 0.001, 0.01, -0.02, 0.015]).astype(np.float32)
    out["T_lin"] = T_lin
    for l in range(c["levels"]):
        out[f"img_l{l}"] = ctx.get_image(0, l)
        out[f"desc_sha_l{l}"] = np.bytes_(";".join(sha(ctx.get_descriptor_channel(1, l, ch)) for ch in range(ctx.Cn)))
        out[f"desc0_l{l}"] = ctx.get_descriptor_channel(1, l, 0)
        out[f"saliency_l{l}"] = ctx.get_saliency(0, l)
        out[f"inds_l{l}"] = ctx.get_point_indices(0, l)
        out[f"points_l{l}"] = ctx.get_points(0, l)
        Tn, Tni = ctx.get_normalization(0, l)
        out[f"norm_l{l}"] = np.stack([Tn, Tni])
        out[f"pixels_l{l}"] = ctx.get_pixels(0, l)
        out[f"jac_l{l}"] = ctx.get_jacobians(0, l)
        lin = ctx.linearize(0, 0, 1, l, T_lin)
        out[f"valid_l{l}"] = ctx.get_valid(0)
        out[f"resid_l{l}"] = ctx.get_residuals(0)
        out[f"weights_l{l}"] = ctx.get_weights(0)
        out[f"H_l{l}"] = lin["H"]
        out[f"G_l{l}"] = lin["G"]
        out[f"lin_scalars_l{l}"] = np.array([lin["f_norm"], lin["sigma"], lin["num_valid"]], np.float64)
    T, stats, trace = ctx.estimate_pose_trace(0, 0, 1)
    out["T_est"] = T
    out["iters"] = np.array([s["numIterations"] for s in stats], np.int32)
    out["status"] = np.array([s["status"] for s in stats], np.int32)
    out["trace_T"] = trace[:, :16].copy()
    out["trace_level"] = trace[:, 67].astype(np.int32)
    out["T_gt"] = d["T_gt"]
    return out


def main():
    ge.build_oracle()
    orc = capi.Binding(ge.ORACLE_LIB, "bpvo_orc_")
    for c in CASES:
        out = build_case(orc, c)
        path = os.path.join(HERE, c["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
