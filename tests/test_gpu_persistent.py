"""The persistent single-launch Gauss-Newton kernel (small groups: sequential addFrame, a handful of pairs) against the
four-kernel chain it replaces: same device functions, same chunk / tile indices, so everything observable must be equal BIT
FOR BIT — poses, statistics, residuals, valid masks, weights, robust scale — and the kernel must really have run
(bpvo_hip_persistent_counts).  reference path: PoseEstimatorBase::run, bpvo/pose_estimator_base.h:324-407."""
import numpy as np
import pytest

from bpvo_amd import capi, synth
from util import bits_equal, make_params, setup_pair, set_options

pytestmark = pytest.mark.gpu


def run_single(hip, rows, cols, levels, descriptor, loss, **kw):
    ctx, d, _ = setup_pair(hip, rows, cols, levels=levels, descriptor=descriptor, loss=loss, **kw)
    T, st = ctx.estimate_pose(0, 0, 1)
    rec = dict(T=T, st=st, frac=ctx.fraction_good(0, 0.85), r=ctx.get_residuals(0), v=ctx.get_valid(0), w=ctx.get_weights(0),
               fused=ctx.fused_point_counts(), med=ctx.median_path_counts(), taps=ctx.tap_cache_counts(), lin=ctx.total_linearizations())
    # a second estimate from the first one's pose (a warm start: few iterations, early exits)
    T2, st2 = ctx.estimate_pose(0, 0, 1, T)
    rec["T2"], rec["st2"], rec["pk"] = T2, st2, ctx.persistent_counts()
    return rec


@pytest.mark.parametrize("rows,cols,levels", [pytest.param(120, 160, 3, id="160x120-L3"), pytest.param(376, 1241, 4, id="kitti-1241x376-L4")])
@pytest.mark.parametrize("descriptor,loss", [("bitplanes", "tukey"), ("bitplanes", "huber"), ("bitplanes", "l2"), ("intensity", "huber"),
                                             ("intensity", "tukey")])
def test_persistent_kernel_is_bit_identical_to_the_chain(hip, rows, cols, levels, descriptor, loss, monkeypatch):
    out = []
    for on in ("0", "1"):
        set_options(monkeypatch, persistent=on)
        out.append(run_single(hip, rows, cols, levels, descriptor, loss))
    a, b = out
    assert a["pk"] == (0, 0)
    assert b["pk"][0] >= 2 * levels and b["pk"][1] == 0, b["pk"]        # every level of both estimates, never gave up
    assert bits_equal(a["T"], b["T"]) and bits_equal(a["T2"], b["T2"])
    assert a["st"] == b["st"] and a["st2"] == b["st2"]
    assert a["frac"] == b["frac"]
    assert np.array_equal(a["v"], b["v"]) and bits_equal(a["r"], b["r"]) and bits_equal(a["w"], b["w"])
    # the same work was done: linearisations, median selections by path, fused points, tap-cache lookups and hits
    assert a["lin"] == b["lin"] and a["med"] == b["med"] and a["fused"] == b["fused"] and a["taps"] == b["taps"]


@pytest.mark.parametrize("n_pairs", [2, 5, 8])
def test_persistent_kernel_groups_of_pairs(hip, n_pairs, monkeypatch):
    """Groups of up to 8 pairs in one persistent launch (option persist_max_ws): pairs converge after different numbers of
    iterations, so workspaces drop out of the loop one by one while the others go on."""
    rows, cols, levels = 120, 160, 3
    b = synth.make_batch(rows, cols, n_pairs, first_index=11)
    out = []
    for on in ("0", "1"):
        set_options(monkeypatch, persistent=on)
        set_options(monkeypatch, persist_max_ws="8")
        ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, descriptor="bitplanes", loss="tukey", levels=levels),
                         n_frames=2 * n_pairs, n_pairs=n_pairs)
        poses, stats = ctx.batch_run(b["images"], b["disparities"])
        out.append(dict(poses=poses, stats=stats, r=ctx.get_residuals(n_pairs - 1), w=ctx.get_weights(n_pairs - 1), pk=ctx.persistent_counts(),
                        lin=ctx.total_linearizations()))
    a, c = out
    assert a["pk"] == (0, 0) and c["pk"] == (levels, 0)
    assert bits_equal(a["poses"], c["poses"]) and a["stats"].tobytes() == c["stats"].tobytes()
    assert bits_equal(a["r"], c["r"]) and bits_equal(a["w"], c["w"])
    assert a["lin"] == c["lin"]
    if n_pairs > 2:
        assert len(np.unique(a["stats"]["numIterations"][:, 0])) > 1          # the pairs really finish at different iterations


def test_persistent_kernel_small_grid_and_sequence(hip, monkeypatch):
    """A grid of ONE workgroup (every chunk and tile looped over by the same four virtual blocks) and of 3 (ragged split), and a
    short addFrame sequence (key-framing, warm starts) through the persistent path."""
    rows, cols, levels = 120, 160, 3
    ref = None
    for on, grid in (("0", "64"), ("1", "1"), ("1", "3"), ("1", "64")):
        set_options(monkeypatch, persistent=on)
        set_options(monkeypatch, persist_grid=grid)
        rec = run_single(hip, rows, cols, levels, "bitplanes", "tukey")
        seq = synth.make_sequence(rows, cols, 6)
        vo = hip.create(seq["K"], seq["b"], rows, cols, make_params(hip, descriptor="bitplanes", loss="tukey", levels=levels), n_frames=3, n_pairs=1)
        rec["traj"] = np.stack([vo.add_frame(img, disp)["pose"] for img, disp in seq["frames"]])
        if ref is None:
            ref = rec
            continue
        assert rec["pk"][1] == 0 and rec["pk"][0] > 0
        assert bits_equal(ref["T"], rec["T"]) and ref["st"] == rec["st"] and bits_equal(ref["r"], rec["r"]) and bits_equal(ref["w"], rec["w"])
        assert bits_equal(ref["traj"], rec["traj"])


def test_persistent_kernel_gives_up_cleanly(hip, monkeypatch):
    """A grid barrier that cannot complete in time (here: a 10 ns budget) makes every workgroup leave without writing the states
    back; the library reruns the group through the four-kernel chain and stays on it.  Same results, no hang."""
    rows, cols, levels = 376, 1241, 4
    set_options(monkeypatch, persistent="0")
    ref = run_single(hip, rows, cols, levels, "bitplanes", "tukey")
    set_options(monkeypatch, persistent="1")
    set_options(monkeypatch, persist_timeout_ticks="1")
    rec = run_single(hip, rows, cols, levels, "bitplanes", "tukey")
    assert rec["pk"][1] == 1 and rec["pk"][0] <= levels          # gave up during the first estimate, never tried again
    assert bits_equal(ref["T"], rec["T"]) and bits_equal(ref["T2"], rec["T2"]) and ref["st"] == rec["st"] and ref["st2"] == rec["st2"]
    assert bits_equal(ref["r"], rec["r"]) and bits_equal(ref["w"], rec["w"]) and np.array_equal(ref["v"], rec["v"])


# ---- the team-persistent kernel for small batches (gn_team_kernel): one launch for the whole Gauss-Newton stage of a batch -----------
def run_batch(hip, rows, cols, levels, n, descriptor, loss, first_index=200, **kw):
    b = synth.make_batch(rows, cols, n, first_index=first_index, workers=min(8, n))
    ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, descriptor=descriptor, loss=loss, levels=levels, **kw), n_frames=2 * n, n_pairs=n)
    poses, stats = ctx.batch_run(b["images"], b["disparities"])
    last = n - 1
    rec = dict(poses=poses, stats=stats, r=ctx.get_residuals(last), v=ctx.get_valid(last), w=ctx.get_weights(last), frac=ctx.fraction_good(0, 0.85),
               lin=ctx.total_linearizations(), med=ctx.median_path_counts(), fused=ctx.fused_point_counts(), taps=ctx.tap_cache_counts())
    # a second batch on the same context (states, tap caches and counters of the first one are in place)
    rec["poses2"], rec["stats2"] = ctx.batch_run(b["images"][::-1].copy(), b["disparities"][::-1].copy())
    rec["team"], rec["pk"] = ctx.team_counts(), ctx.persistent_counts()
    rec["joins"] = ctx.get_option("team_joins_seen")
    ctx.close()
    return rec


def assert_same_batch(a, b):
    assert bits_equal(a["poses"], b["poses"]) and a["stats"].tobytes() == b["stats"].tobytes()
    assert bits_equal(a["poses2"], b["poses2"]) and a["stats2"].tobytes() == b["stats2"].tobytes()
    assert bits_equal(a["r"], b["r"]) and bits_equal(a["w"], b["w"]) and np.array_equal(a["v"], b["v"]) and a["frac"] == b["frac"]
    assert a["lin"] == b["lin"] and a["med"] == b["med"] and a["fused"] == b["fused"] and a["taps"] == b["taps"]


@pytest.mark.parametrize("rows,cols,levels,n", [pytest.param(120, 160, 3, 5, id="160x120-5pairs"), pytest.param(120, 160, 3, 37, id="160x120-37pairs"),
                                                pytest.param(376, 1241, 4, 12, id="kitti-12pairs")])
@pytest.mark.parametrize("descriptor,loss", [("bitplanes", "tukey"), ("bitplanes", "l2"), ("intensity", "huber")])
def test_team_kernel_is_bit_identical_to_the_chain(hip, rows, cols, levels, n, descriptor, loss, monkeypatch):
    """Batches of 2 .. 256 pairs: every pair through all its levels in ONE launch, a team of workgroups per pair, against the
    four-kernel chain with its host-driven level loop — poses, statistics, residuals, valid masks, weights, robust scale, and the
    counters of the work done (linearisations, median selections by path, fused points, tap-cache lookups and hits)."""
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, descriptor, loss)
    assert ref["team"] == 0
    set_options(monkeypatch, team="1")
    got = run_batch(hip, rows, cols, levels, n, descriptor, loss)
    assert got["team"] == 2 and got["pk"][1] == 0, (got["team"], got["pk"])
    assert_same_batch(ref, got)
    assert len(np.unique(ref["stats"]["numIterations"][:, 0])) > 1            # the pairs really finish at different iterations


@pytest.mark.parametrize("cus,team_size", [(6, 0), (7, 3), (4, 4), (200, 0)])
def test_team_kernel_shapes(hip, cus, team_size, monkeypatch):
    """Fewer teams than pairs (pairs are handed out dynamically: option team_cus caps the grid), teams of 1, 3 and 4 workgroups
    (ragged chunk / tile splits), more CUs claimed than pairs need."""
    rows, cols, levels, n = 120, 160, 3, 13
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=777)
    set_options(monkeypatch, team="1")
    set_options(monkeypatch, team_cus=str(cus))
    if team_size:
        set_options(monkeypatch, team_size=str(team_size))
    got = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=777)
    assert got["team"] == 2 and got["pk"][1] == 0
    assert_same_batch(ref, got)


@pytest.mark.parametrize("rows,cols,levels,n,cus,join", [pytest.param(120, 160, 3, 24, 48, 2, id="24-pairs-teams-of-2-any-team"),
                                                        pytest.param(120, 160, 3, 24, 48, 1, id="24-pairs-teams-of-2-own-xcd"),
                                                        pytest.param(120, 160, 3, 13, 39, 2, id="13-pairs-teams-of-3-across-xcds"),
                                                        pytest.param(120, 160, 3, 12, 32, 2, id="12-pairs-teams-of-2-and-8-spare-workgroups"),
                                                        pytest.param(120, 160, 3, 16, 40, 1, id="16-pairs-teams-of-2-and-8-spares-own-xcd"),      # (own-XCD joins need teams that sit on one XCD: a multiple of 8 teams)
                                                        pytest.param(376, 1241, 4, 96, 0, 2, id="kitti-96-pairs-teams-of-2-and-64-spares"),
                                                        pytest.param(376, 1241, 4, 64, 0, 2, id="kitti-64-pairs-teams-of-4"),
                                                        pytest.param(480, 640, 4, 40, 0, 2, id="640x480-40-pairs-teams-of-6")])
@pytest.mark.parametrize("descriptor,loss", [("bitplanes", "tukey"), ("intensity", "huber")])
def test_teams_that_grow_are_bit_identical_to_the_chain(hip, rows, cols, levels, n, cus, join, descriptor, loss, monkeypatch):
    """Workgroups of a team that has run out of pairs join the teams still at work (kernels_gn_team.hip, pk_join_team): the batch against the
    four-kernel chain — poses, statistics, residuals, valid masks, weights, counters — with joins actually taking place (counter
    team_joins_seen), on one XCD only and across XCDs (a newcomer from another XCD turns the team's barriers into agent-scope ones)."""
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, descriptor, loss, first_index=400)
    assert ref["team"] == 0
    set_options(monkeypatch, team="1", team_join=str(join), team_join_from_pairs="0", team_max_pairs="128", team_full_pairs="128")
    if cus:
        set_options(monkeypatch, team_cus=str(cus))
    got = run_batch(hip, rows, cols, levels, n, descriptor, loss, first_index=400)
    assert got["team"] == 2 and got["pk"][1] == 0, (got["team"], got["pk"])
    assert got["joins"] > 0, "no workgroup joined another team: the case does not test what it says"
    assert_same_batch(ref, got)


def test_growing_teams_give_up_cleanly(hip, monkeypatch):
    """The growing form of the team kernel with a barrier budget of 10 ns: barriers and the newcomers' waiting loops all watch the abort word,
    every workgroup leaves, the library reruns the batch through the chain and stays on it.  Same results, no hang."""
    rows, cols, levels, n = 120, 160, 3, 24
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    set_options(monkeypatch, team="1", team_join="2", team_join_from_pairs="0", team_cus="48", persist_timeout_ticks="1")
    got = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    assert got["pk"][1] == 1 and got["team"] == 1          # launched once, gave up, never tried again
    assert bits_equal(ref["poses"], got["poses"]) and ref["stats"].tobytes() == got["stats"].tobytes()
    assert bits_equal(ref["poses2"], got["poses2"]) and bits_equal(ref["r"], got["r"]) and bits_equal(ref["w"], got["w"])


def test_team_cus_above_the_device_is_refused_and_persistent_rearms(hip, monkeypatch):
    """Option team_cus is clamped by refusal (a grid larger than the device cannot be co-resident and would only time out); and
    set_option("persistent", 1) re-arms a context whose team launch once gave up (persistent_failed is sticky otherwise)."""
    rows, cols, levels, n = 120, 160, 3, 9
    b = synth.make_batch(rows, cols, n, first_index=31)
    ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, descriptor="bitplanes", loss="tukey", levels=levels), n_frames=2 * n, n_pairs=n)
    with pytest.raises(capi.BpvoError):
        ctx.set_option("team_cus", 100000)
    assert ctx.get_option("team_cus") <= 1024
    ctx.set_option("team_size", 4)
    ctx.set_option("persist_timeout_ticks", 1)
    ctx.batch_run(b["images"], b["disparities"])
    assert ctx.persistent_counts()[1] == 1            # gave up at the first barrier, ran on the chain
    ctx.set_option("persist_timeout_ticks", 50000000)
    ctx.set_option("persistent", 1)
    assert ctx.persistent_counts()[1] == 0
    before = ctx.team_counts()
    ctx.batch_run(b["images"], b["disparities"])
    assert ctx.persistent_counts()[1] == 0 and ctx.team_counts() > before
    ctx.close()


def test_team_kernel_gives_up_cleanly(hip, monkeypatch):
    """A team barrier that cannot complete in its budget (10 ns): every workgroup leaves, the library reruns the batch through the chain
    and stays on it.  Same results, no hang."""
    rows, cols, levels, n = 120, 160, 3, 9
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    set_options(monkeypatch, team="1")
    set_options(monkeypatch, team_size="4")
    set_options(monkeypatch, persist_timeout_ticks="1")
    got = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    assert got["pk"][1] == 1 and got["team"] == 1          # launched once, gave up, never tried again
    assert bits_equal(ref["poses"], got["poses"]) and ref["stats"].tobytes() == got["stats"].tobytes()
    assert bits_equal(ref["poses2"], got["poses2"]) and bits_equal(ref["r"], got["r"]) and bits_equal(ref["w"], got["w"])


def test_split_team_launch_gives_up_cleanly(hip, monkeypatch):
    """Batches of up to team_split_max_pairs pairs run the team kernel in TWO launches (the coarsest level, then the others behind the deferred
    normalisation).  A first launch that gives up at a barrier (budget 10 ns) hands its abort word on: the second launch leaves in its prologue
    instead of running every remaining level on states that were never written back; the library reruns the batch on the chain.  Same results."""
    rows, cols, levels, n = 120, 160, 3, 3
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    set_options(monkeypatch, team="1", team_split_max_pairs="4", persist_timeout_ticks="1")
    got = run_batch(hip, rows, cols, levels, n, "bitplanes", "tukey", first_index=31)
    assert got["pk"][1] == 1 and got["team"] == 1          # launched once (two launches of one run), gave up, never tried again
    assert bits_equal(ref["poses"], got["poses"]) and ref["stats"].tobytes() == got["stats"].tobytes()
    assert bits_equal(ref["poses2"], got["poses2"]) and bits_equal(ref["r"], got["r"]) and bits_equal(ref["w"], got["w"])


@pytest.mark.parametrize("n", [16, 40])
@pytest.mark.parametrize("descriptor,loss", [("bitplanes", "tukey"), ("intensity", "huber")])
def test_team_kernel_with_a_team_per_xcd(hip, n, descriptor, loss, monkeypatch):
    """Team counts that divide over the eight XCDs: a team's workgroups are dealt to ONE XCD, every workgroup registers the XCD it runs on
    (HW_REG_XCC_ID), and a team that finds itself on one XCD drops the L2 write-back of its barriers (option team_local_barriers; its
    barrier between reduction and step carries the partials past the caches either way).  Against the agent-scope fences and against the
    chain, bit for bit — with one channel too (whose plain residual loads read stale lines under a weaker invalidation)."""
    rows, cols, levels = 120, 160, 3
    set_options(monkeypatch, team="0")
    ref = run_batch(hip, rows, cols, levels, n, descriptor, loss, first_index=310)
    assert ref["team"] == 0
    set_options(monkeypatch, team="1")
    for local in ("0", "1"):
        set_options(monkeypatch, team_local_barriers=local)
        got = run_batch(hip, rows, cols, levels, n, descriptor, loss, first_index=310)
        assert got["team"] == 2 and got["pk"][1] == 0, (local, got["team"], got["pk"])
        assert_same_batch(ref, got)


def test_which_batches_take_the_team_kernel(hip):
    """Up to 128 pairs (team_max_pairs) a batch takes the team kernel: since round 5 the grid always fills the chip — what the division CUs /
    pairs leaves over starts as spare workgroups that join the teams (96 pairs: 2 x 96 + 64 spares).  With the spares (or the growing form)
    switched off the round-4 rule holds: above 80 pairs (team_full_pairs) only when CUs / pairs workgroups per pair use at least 95 % of the
    CUs — on 256 CUs 128 pairs (2 x 128) and 85 (3 x 85) do, 96 (2 x 96) do not."""
    rows, cols, levels = 96, 128, 2
    for n, team in ((128, True), (96, False), (85, True), (129, False)):
        b = synth.make_batch(rows, cols, n, first_index=5, workers=8)
        for spares in (1, 0):
            ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, levels=levels), n_frames=2 * n, n_pairs=n)
            ctx.set_option("team_spares", spares)
            cus = int(ctx.get_option("team_cus"))
            ctx.batch_run(b["images"], b["disparities"])
            ts = max(1, min(64, cus // n))
            fills = 20 * ts * min(n, cus // ts) >= 19 * cus
            want = (n <= 128) if spares else (n <= 128 and fills)
            assert (ctx.team_counts() == 1) == want, (n, spares, cus, ctx.team_counts())
            if cus == 256 and not spares:
                assert fills == team or n > 128, (n, fills)
            if spares and n == 96 and cus == 256:
                assert ctx.get_option("team_joins_seen") >= 64      # the 64 spare workgroups found a team
            ctx.close()


def test_dense_templates_take_the_chain_not_the_team_kernel(hip, monkeypatch):
    """A small batch whose templates have a pyramid level of more points than the persistent kernels are given (option persist_max_points: NMS
    off on a large level, conf/tsukuba.cfg) is bandwidth work for the chain's chip-wide launches: the team kernel's per-pair median walks a
    thousand candidate segments per iteration there (4 dense 640 x 480 pairs: 11.5 ms per step on the team kernel, 6.0 on the chain:
    scripts/dense_batch_ab.py).  Same bits either way; with the threshold raised the team kernel takes the batch again."""
    rows, cols, levels, n = 480, 640, 3, 4
    b = synth.make_batch(rows, cols, n, first_index=40)
    kw = dict(levels=levels, descriptor="bitplanes", loss="tukey", nonMaxSuppRadius=0)
    outs = []
    for raise_threshold in (False, True):
        ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, **kw), n_frames=2 * n, n_pairs=n)
        if raise_threshold:
            ctx.set_option("persist_max_points", 1 << 22)
        outs.append(ctx.batch_run(b["images"], b["disparities"]))
        assert max(ctx.num_points(0, l) for l in range(levels)) > 32768
        assert ctx.team_counts() == (1 if raise_threshold else 0), (raise_threshold, ctx.team_counts())
        ctx.close()
    assert bits_equal(outs[0][0], outs[1][0]) and outs[0][1].tobytes() == outs[1][1].tobytes()
    # templates with non-maximum suppression: the team kernel, as before
    ctx = hip.create(b["K"], b["b"], rows, cols, make_params(hip, levels=levels, descriptor="bitplanes", loss="tukey"), n_frames=2 * n, n_pairs=n)
    ctx.batch_run(b["images"], b["disparities"])
    assert ctx.team_counts() == 1
    ctx.close()


@pytest.mark.parametrize("rows,cols,levels,descriptor,loss", [pytest.param(376, 1241, 4, "bitplanes", "tukey", id="kitti-bitplanes"),
                                                              pytest.param(480, 640, 5, "bitplanes", "huber", id="vga-L5"),
                                                              pytest.param(120, 160, 3, "intensity", "huber", id="160x120-intensity"),
                                                              pytest.param(97, 133, 2, "bitplanes", "tukey", id="97x133-L2")])
def test_one_pair_per_call_is_bit_identical_whatever_ran_before(hip, rows, cols, levels, descriptor, loss, monkeypatch):
    """bpvo_hip_batch_run with ONE pair per call — the path with the frame stage's levels in merged launches, the one-kernel pyramid, the
    normalisation of every level but the coarsest still running on the side stream when the first iterations start, the level starts folded
    into the persistent kernel, table + poses in one launch — on ONE context fed six DIFFERENT pairs in a row, against each pair
    estimated through setData / setTemplate / estimatePose on a fresh context: poses and statistics bit for bit.  A normalisation (or a
    tap-cache key, or a state word) left over from the pair before would show here; with the options off the same must hold."""
    n = 6
    b = synth.make_batch(rows, cols, n, first_index=40)
    p = make_params(hip, levels=levels, descriptor=descriptor, loss=loss)
    want = []
    for k in range(n):
        sc = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
        sc.frame_set_data(0, b["images"][2 * k], b["disparities"][2 * k])
        sc.frame_set_template(0)
        sc.frame_set_data(1, b["images"][2 * k + 1], b["disparities"][2 * k + 1])
        want.append(sc.estimate_pose(0, 0, 1))
        sc.close()
    for opts in (dict(), dict(normalization_deferred=0), dict(levels_in_one_launch_max_frames=0, small_batch_fused=0, normalization_side_stream=0)):
        set_options(monkeypatch, **opts)
        ctx = hip.create(b["K"], b["b"], rows, cols, p, n_frames=2, n_pairs=1)
        for rep in range(2):
            for k in range(n):
                poses, stats = ctx.batch_run(b["images"][2 * k: 2 * k + 2], b["disparities"][2 * k: 2 * k + 2])
                T, st = want[k]
                assert bits_equal(T, poses[0]), (opts, rep, k)
                for l in range(levels):
                    assert st[l]["numIterations"] == int(stats["numIterations"][0, l]) and st[l]["status"] == int(stats["status"][0, l]), (opts, rep, k, l)
        if not opts:
            assert ctx.persistent_counts()[0] >= 2 * n and ctx.persistent_counts()[1] == 0, ctx.persistent_counts()      # the persistent kernel ran, never gave up
        ctx.close()
