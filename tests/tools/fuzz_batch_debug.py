"""Debug: the batch-vs-single check of tests/test_gpu_fuzz.py on every fifth normalised draw of a seed, with a diagnosis when it fails."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd
from bpvo_amd import capi, synth
import fuzz_parity as fz
from util import make_params, bits_equal


def diagnose(hip, rows, cols, kw, seed, opts):
    kw = dict(kw)
    fast_warp, fuse = kw.pop("_fast_warp", False), kw.pop("_fuse_frozen", False)
    kw.pop("_dspace", False)
    n = 2 + seed % 4
    b = synth.make_batch(rows, cols, n, first_index=seed % 3000)
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0") + opts
    bc = hip.create(b["K"], b["b"], rows, cols, make_params(hip, **kw), n_frames=2 * n, n_pairs=n)
    poses, stats = bc.batch_run(b["images"], b["disparities"])
    os.environ["BPVO_HIP_OPTIONS"] = "fuse_frozen=" + ("1" if fuse else "0")
    sc = hip.create(b["K"], b["b"], rows, cols, make_params(hip, **kw), n_frames=2, n_pairs=1)
    res = []
    for k in range(n):
        sc.frame_set_data(0, b["images"][2 * k], b["disparities"][2 * k])
        sc.frame_set_template(0)
        sc.frame_set_data(1, b["images"][2 * k + 1], b["disparities"][2 * k + 1])
        try:
            T, st = sc.estimate_pose(0, 0, 1)
        except capi.BpvoError:
            res.append("err")
            continue
        npts_b = [bc.num_points(2 * k, l) for l in range(kw["levels"])]
        npts_s = [sc.num_points(0, l) for l in range(kw["levels"])]
        same_idx = all(np.array_equal(bc.get_point_indices(2 * k, l), sc.get_point_indices(0, l)) for l in range(kw["levels"]))
        same_pix = all(bits_equal(bc.get_pixels(2 * k, l), sc.get_pixels(0, l)) for l in range(kw["levels"]))
        sal_b, sal_s = bc.get_saliency(2 * k, 0), sc.get_saliency(0, 0)
        dif = np.argwhere(sal_b.view(np.uint32) != sal_s.view(np.uint32))
        where = None
        if len(dif):
            where = dict(n=len(dif), rows=(int(dif[:, 0].min()), int(dif[:, 0].max())), cols=(int(dif[:, 1].min()), int(dif[:, 1].max())),
                         first=[tuple(int(v) for v in d) for d in dif[:6]], vals=[(float(sal_b[tuple(d)]), float(sal_s[tuple(d)])) for d in dif[:4]])
        res.append(dict(pose=bits_equal(T, poses[k]), npts=(npts_b, npts_s), idx=same_idx, pix=same_pix, sal=where,
                        it=([int(stats["numIterations"][k, l]) for l in range(kw["levels"])], [s["numIterations"] for s in st])))
    bc.close(); sc.close()
    return res


def main():
    hip = bpvo_amd.load()
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 20261002
    rng = np.random.default_rng(seed0)
    n = 0
    while n < 160:
        rows, cols, kw, scene, s = fz.draw(rng)
        if not kw["withNormalization"] or kw.get("_dspace"):
            continue
        n += 1
        if n % 5 or (len(sys.argv) > 2 and n != int(sys.argv[2])):
            continue
        try:
            out = fz.check_batch(hip, rows, cols, kw, s)
        except AssertionError as e:
            print("FAIL case", n, rows, cols, kw, s, e.args)
            for opts in ("", ",lazy_template_descriptor=0", ",upload_workers=0", ",lanes=1"):
                print("  options", repr(opts))
                for k, r in enumerate(diagnose(hip, rows, cols, kw, s, opts)):
                    print("    pair", k, r)
            continue
        print("case", n, out)


if __name__ == "__main__":
    main()
