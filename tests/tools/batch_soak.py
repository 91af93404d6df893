"""Soak of bpvo_hip_batch_run against the single-pair path, bit for bit, on random configurations (tests/tools/fuzz_parity.py draw): every case
with a random set of scheduling options of the batch context (the team kernel or the chain, the step inside the reduction or not, lazy
template levels or dense ones, one or two lanes) and on a context that has run a batch of OTHER images before (stale buffers are then
somebody else's data).  No oracle involved: the two paths of the product against each other.

  python tests/tools/batch_soak.py --seconds 600 --seed 3
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import bpvo_amd
import fuzz_parity as fz

OPTION_SETS = ["", "team=0", "team=0,step_in_reduce_max_pairs=0", "lazy_template_descriptor=0", "team=0,lanes=1", "team=0,upload_workers=0",
               "keep_current_disparity=1", "team=0,lanes=2",
               # teams that grow (round 5): forced on for every batch size, teams of 2 - 4 so that workgroups do run out of pairs and join others;
               # joins on the workgroup's own XCD only, and a grid smaller than the chip
               "team_join_from_pairs=0,team_size=2", "team_join_from_pairs=0,team_size=3,team_join=1", "team_join_from_pairs=0,team_size=4,team_cus=64",
               "team_join_from_pairs=0,team_size=2,team_cus=24",
               # one pair per call and the smallest team batches (round 5): each piece of the rebuilt frame stage / launch sequence switched off
               "normalization_deferred=0", "levels_in_one_launch_max_frames=0", "team_split_max_pairs=0", "small_batch_fused=0,normalization_side_stream=0",
               "team_split_max_pairs=8"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-rows", type=int, default=200)
    ap.add_argument("--max-cols", type=int, default=300)
    a = ap.parse_args()
    fz.MAX_ROWS, fz.MAX_COLS = a.max_rows, a.max_cols
    hip = bpvo_amd.load()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    outcomes, fails, n = {}, 0, 0
    while time.time() - t0 < a.seconds:
        rows, cols, kw, scene, seed = fz.draw(rng)
        if kw.get("_dspace"):
            continue
        n += 1
        opts = OPTION_SETS[int(rng.integers(len(OPTION_SETS)))]
        npairs = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 9, 16, 17, 24, 32, 40]))      # (1: the single-pair path inside batch_run; multiples of 8: teams that sit on one XCD each)
        if npairs * rows * cols > 40 * 120 * 160:
            npairs = max(1 if npairs == 1 else 2, (40 * 120 * 160) // (rows * cols))
        try:
            out = fz.check_batch(hip, rows, cols, kw, seed, options=opts, dirty=True, n=npairs)
        except AssertionError as e:
            out, fails = "FAIL", fails + 1
            print("FAIL", rows, cols, seed, repr(opts), npairs, kw, e.args, flush=True)
        except Exception:
            out, fails = "EXCEPTION", fails + 1
            print("EXCEPTION", rows, cols, seed, repr(opts), npairs, kw, traceback.format_exc(), flush=True)
        outcomes[out] = outcomes.get(out, 0) + 1
    print("cases", n, "seconds", round(time.time() - t0, 1), "outcomes", outcomes, flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
