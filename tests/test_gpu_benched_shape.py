"""The shape bench.py's headline is quoted on, under test: ONE batch of 1024 frame pairs of 1241x376 bit-planes / 4 levels / Tukey on
one GPU (BASELINE.json config 5 at its largest single-GPU form) — two estimation lanes, the 512-thread shape of the median at real
launch widths, and, for host buffers, the upload pipeline with its three-group plan at real chunk sizes.

 * resident inputs and host buffers give the same poses and statistics, bit for bit;
 * 32 pairs spread over the whole batch — both lanes, all three upload groups — against the CPU oracle within the north-star bar
   (1e-4 rad / 1e-3 m; reference path per pair: bpvo/vo_pose_estimator.cc:63-93);
 * pairs are independent, so the first 128 pairs of the batch (seeds 1000 .. 1127 — what rank 0 of the 8-GPU job of config 5 holds)
   must equal a 128-pair shard run on a context of its own, bit for bit: poses, iteration counts, statuses.  The tile partials are
   cut by a rule that depends on the channel count only (gn_pts_per_block), so a pair's rounding cannot depend on the batch it is in.
"""
import os

import numpy as np
import pytest

from bpvo_amd import synth
from util import ROT_TOL, TRANS_TOL, bits_equal, make_params, pose_error

pytestmark = pytest.mark.gpu

ROWS, COLS, LEVELS = 376, 1241, 4
BATCH, SHARD = 1024, 128
PICKS = list(range(5, BATCH, 32))       # 32 pairs: 16 per lane of the resident run; 6 / 16 / 10 in the three groups of the host plan


@pytest.fixture(scope="module")
def benched(hip):
    import torch
    kw = dict(descriptor="bitplanes", loss="tukey", levels=LEVELS)
    batch = synth.make_batch(ROWS, COLS, BATCH, first_index=0, workers=min(32, os.cpu_count() or 1))      # seeds 1000 .. 2023
    # the shard first, on a context of its own (a context only fans out over lanes while it is the only one on the device)
    ctx = hip.create(batch["K"], batch["b"], ROWS, COLS, make_params(hip, **kw), n_frames=2 * SHARD, n_pairs=SHARD)
    shard_poses, shard_stats = ctx.batch_run(batch["images"][: 2 * SHARD], batch["disparities"][: 2 * SHARD])
    ctx.close()
    ctx = hip.create(batch["K"], batch["b"], ROWS, COLS, make_params(hip, **kw), n_frames=2 * BATCH, n_pairs=BATCH)
    dev = torch.device("cuda", 0)
    di = torch.from_numpy(batch["images"]).to(dev)
    dd = torch.from_numpy(batch["disparities"]).to(dev)
    res_poses, res_stats = ctx.batch_run_device(BATCH, di.data_ptr(), dd.data_ptr())
    del di, dd
    torch.cuda.empty_cache()
    host_poses, host_stats = ctx.batch_run(batch["images"], batch["disparities"])
    up_s, up_bytes = ctx.upload_stats()
    ctx.close()
    return dict(batch=batch, kw=kw, shard=(shard_poses, shard_stats), resident=(res_poses, res_stats), host=(host_poses, host_stats),
                upload=(up_s, up_bytes))


def test_resident_and_host_buffer_batches_are_bit_identical(benched):
    (rp, rs), (hp, hs) = benched["resident"], benched["host"]
    assert bits_equal(rp, hp) and rs.tobytes() == hs.tobytes()
    up_s, up_bytes = benched["upload"]
    assert up_bytes >= 2 * BATCH * ROWS * COLS          # the pipeline really carried the batch: two u8 images per pair at the very least
    print(f"\nhost-buffer batch: {up_bytes / 1e9:.2f} GB staged in {1e3 * up_s:.1f} ms")


def test_the_shard_inside_the_batch_equals_the_shard_alone(benched):
    (bp, bs), (sp, ss) = benched["resident"], benched["shard"]
    assert bits_equal(bp[:SHARD], sp)
    assert bs[:SHARD].tobytes() == ss.tobytes()
    assert len(np.unique(bs["numIterations"][:, 0])) > 1


def test_benched_batch_against_the_cpu_path(benched, orc):
    batch, kw = benched["batch"], benched["kw"]
    poses, stats = benched["resident"]
    ctx = orc.create(batch["K"], batch["b"], ROWS, COLS, make_params(orc, **kw), n_frames=2, n_pairs=1)
    worst = (0.0, 0.0)
    for k in PICKS:
        ctx.frame_set_data(0, batch["images"][2 * k], batch["disparities"][2 * k])
        ctx.frame_set_template(0)
        ctx.frame_set_data(1, batch["images"][2 * k + 1], batch["disparities"][2 * k + 1])
        T, st = ctx.estimate_pose(0, 0, 1)
        rot, tr = pose_error(poses[k], T)
        assert rot <= ROT_TOL and tr <= TRANS_TOL, (k, rot, tr, stats["numIterations"][k].tolist(), [s["numIterations"] for s in st])
        worst = (max(worst[0], rot), max(worst[1], tr))
    ctx.close()
    print(f"\n1024-pair batch: worst pose disagreement over {len(PICKS)} pairs spread over both lanes: {worst[0]:.2e} rad, {worst[1]:.2e} m")
    # the whole batch: rigid transforms, accuracy against the scenes' ground truth
    R = poses[:, :3, :3].astype(np.float64)
    assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-4
    dt = np.linalg.norm(poses[:, :3, 3] - batch["T_gt"][:, :3, 3], axis=1)
    assert np.median(dt) < 5e-3 and dt.max() < 5e-2, (np.median(dt), dt.max())
