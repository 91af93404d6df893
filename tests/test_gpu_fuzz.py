"""A bounded run of the randomised stage-by-stage parity tool (tests/tools/fuzz_parity.py) inside the suite: fixed seeds, the
configurations the reference can be configured to — every descriptor on the device path, the four interpolation types, CD3 / CD5, three
losses, NMS on / off, ragged sizes, 1-5 levels, maxIterations 50 / 100 / 400, the three warp formulations, the fused path, and, for a
third of the cases, the UN-NORMALISED class: withNormalization = 0 (conf/tsukuba_eval.cfg:8 — the default configuration of
apps/eval_descriptors.cc:130, the app that runs every descriptor of the factory) and the DisparitySpaceWarp formulation, whose
setNormalization is a no-op (bpvo/disparity_space_warp.h:87-90).  Every stage up to the weights bit-identical with the oracle, every
final pose inside the bar or explained by one of the tool's rules — each of which asks the ORACLE ITSELF (its other summation orders, a
perturbed start, perturbed normal equations, its f64 trace) whether the difference is the problem's, not the implementation's.  Plus
the committed regression cases, replayed."""
import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))

pytestmark = pytest.mark.gpu

# Outcomes of fuzz_parity.check.  AGREED: both sides behave identically and there is no pose to compare (the same error raised by the
# product and by the oracle — an empty template, a solver failure — or the same non-finite pose): parity of the error path.  EXPLAINED: every
# stage up to the weights is bit-identical, the final pose is outside the bar, and one of the tool's rules shows — by asking the oracle
# itself — that the difference is the problem's.  Each explained rule is capped, and so is their sum: a rule that starts to fire often is a
# finding, not an explanation.
AGREED = {"template-error", "estimate-error", "non-finite"}
EXPLAINED = {"unstable-problem", "iteration-limit", "stops-where-the-oracle-would", "function-tol-at-the-noise-floor", "genuine-function-tol-stop",
             "genuine-scale-freeze", "noise-floor-minimum", "solver-fallback-edge", "solver-acceptance-flip"}
ACCEPTED = {"ok"} | AGREED | EXPLAINED
RULE_CAP, EXPLAINED_CAP, AGREED_CAP = 0.03, 0.08, 0.05      # fractions of the cases of a run
UNNORMALISED_SHARE = 1.0 / 3.0                              # of the cases of a run
# (case, level) cells whose numIterations AND status equal the oracle's under the reference's timing tolerances (conf/perf_*.cfg):
# calibrated against the reference's OWN spread: the same cells for the oracle's 8-chunk reduction (the reference's TBB build,
# linear_system_builder.cc:233-237) and its f64 accumulation against its serial f32 run.  The GPU's agreement with the serial oracle may not be
# worse than the worse of those two by more than ITERATION_CELLS_SLACK (profiles/r06_fuzz_outcomes.txt); an absolute floor stays as a backstop.
ITERATION_CELLS_SLACK = 0.05
ITERATION_CELLS_FLOOR, ITERATION_CELLS_WITHIN_ONE_FLOOR = 0.60, 0.75


@pytest.mark.parametrize("seed,n_cases", [(20261001, 165), (20261002, 165)])
def test_bounded_fuzz_with_the_unnormalised_class(hip, orc, seed, n_cases):
    import fuzz_parity as fz
    rng = np.random.default_rng(seed)
    quota = {True: int(round(UNNORMALISED_SHARE * n_cases)), False: n_cases - int(round(UNNORMALISED_SHARE * n_cases))}
    taken = {True: 0, False: 0}
    outcomes, by_class = {}, {True: {}, False: {}}
    cells = [0, 0, 0]
    own = [0, 0, 0, 0]      # the oracle's 8-chunk / f64 runs against its serial run: equal, within one (each)
    n = 0
    while n < n_cases:
        rows, cols, kw, scene, s = fz.draw(rng)
        un = fz.is_unnormalised(kw)
        if taken[un] >= quota[un]:
            continue
        taken[un] += 1
        n += 1
        out = fz.check(hip, orc, rows, cols, kw, scene, s)        # raises AssertionError with the stage that differs
        assert out in ACCEPTED, (rows, cols, scene, s, kw, out)
        outcomes[out] = outcomes.get(out, 0) + 1
        by_class[un][out] = by_class[un].get(out, 0) + 1
        if n % 5 == 0 and out == "ok":
            outb = fz.check_batch(hip, rows, cols, kw, s, dirty=True)      # (the batch context has run other images before)
            outcomes["batch-" + outb] = outcomes.get("batch-" + outb, 0) + 1
        if n % 3 == 0 and out == "ok":
            e, c1, t, e8, c8, e64, c64 = fz.iteration_cells(hip, orc, rows, cols, kw, scene, s, calibrate=True)
            cells[0] += e; cells[1] += t; cells[2] += c1
            own[0] += e8; own[1] += c8; own[2] += e64; own[3] += c64
    frac, frac1 = cells[0] / max(1, cells[1]), cells[2] / max(1, cells[1])
    own_f = [v / max(1, cells[1]) for v in own]
    table = (f"fuzz seed {seed}: {n} cases ({taken[True]} un-normalised), outcomes {dict(sorted(outcomes.items()))}; "
             f"normalised {dict(sorted(by_class[False].items()))}; un-normalised {dict(sorted(by_class[True].items()))}; "
             f"iteration cells equal under the timing tolerances {cells[0]}/{cells[1]} = {frac:.4f}, numIterations within one {cells[2]}/{cells[1]} = {frac1:.4f}; "
             f"the oracle's own spread over the same cells: 8-chunk reduction equal {own[0]}/{cells[1]} = {own_f[0]:.4f}, within one {own_f[1]:.4f}; "
             f"f64 accumulation equal {own[2]}/{cells[1]} = {own_f[2]:.4f}, within one {own_f[3]:.4f}")
    print("\n" + table)
    out_dir = os.path.join(ROOT, "gpurun_out")       # (scratch that travels back from the GPU box; the committed copy: profiles/r05_fuzz_outcomes.txt)
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "fuzz_outcomes.txt"), "a") as f:
        f.write(table + "\n")
    explained = {k: v for k, v in outcomes.items() if k in EXPLAINED}
    assert all(v <= RULE_CAP * n_cases for v in explained.values()), ("a rule explains more than %g of the cases" % RULE_CAP, outcomes)
    assert sum(explained.values()) <= EXPLAINED_CAP * n_cases, outcomes
    assert sum(v for k, v in outcomes.items() if k in AGREED) <= AGREED_CAP * n_cases, outcomes
    assert outcomes.get("ok", 0) >= (1.0 - EXPLAINED_CAP - AGREED_CAP) * n_cases, outcomes
    # the un-normalised class on its own: most of its poses are inside the bar, the rest explained by the capped rules above
    assert by_class[True].get("ok", 0) >= 0.80 * taken[True], by_class[True]
    assert not [k for k in outcomes if k.startswith("batch-") and k != "batch-ok"], outcomes
    assert cells[1] >= 40 and frac >= ITERATION_CELLS_FLOOR and frac1 >= ITERATION_CELLS_WITHIN_ONE_FLOOR, (cells, frac, frac1)
    assert frac >= min(own_f[0], own_f[2]) - ITERATION_CELLS_SLACK and frac1 >= min(own_f[1], own_f[3]) - ITERATION_CELLS_SLACK, (cells, own, frac, frac1, own_f)


@pytest.mark.parametrize("seed,n_cases", [(20261001, 165), (20261002, 165)])
def test_bounded_fuzz_in_reference_order_is_bit_exact(hip, orc, seed, n_cases):
    """The same 330 draws with the library's validation mode "reference_reduction" (kernels_gn_ref.hip: the reference's f32 index-order sums):
    no rule may fire, no tolerance applies — every case is 'ok-bit-exact' (every linearisation's pose / H / G / f / sigma / valid count / step,
    the final pose, numIterations and status of every level equal to the oracle's bit for bit) or raises on both sides."""
    import fuzz_parity as fz
    rng = np.random.default_rng(seed)
    quota = {True: int(round(UNNORMALISED_SHARE * n_cases)), False: n_cases - int(round(UNNORMALISED_SHARE * n_cases))}
    taken = {True: 0, False: 0}
    outcomes = {}
    n = 0
    while n < n_cases:
        rows, cols, kw, scene, s = fz.draw(rng)
        un = fz.is_unnormalised(kw)
        if taken[un] >= quota[un]:
            continue
        taken[un] += 1
        n += 1
        out = fz.check_reference_order(hip, orc, rows, cols, kw, scene, s)      # raises AssertionError naming the linearisation and field
        assert out in fz.REFERENCE_ORDER_OUTCOMES, (rows, cols, scene, s, kw, out)
        outcomes[out] = outcomes.get(out, 0) + 1
    table = f"fuzz seed {seed} in reference order: {n} cases ({taken[True]} un-normalised), outcomes {dict(sorted(outcomes.items()))}"
    print("\n" + table)
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "fuzz_outcomes.txt"), "a") as f:
        f.write(table + "\n")
    assert outcomes.get("ok-bit-exact", 0) >= (1.0 - AGREED_CAP) * n_cases, outcomes


def test_the_unnormalised_regression_cases_are_explained(hip, orc):
    """tests/tools/fuzz_regressions.txt: the un-normalised cases (withNormalization = 0 or the d-space warp) that rounds 2-4 listed as
    'beyond the bar, no rule': every one is now inside the bar or explained by a rule that asks the oracle — most by
    'solver-fallback-edge' (the oracle's own final pose moves beyond the bar under 2e-7 perturbations of its normal equations and its
    other summation orders, and the GPU's pose lies within 3x that spread or is a fixed point of the oracle)."""
    import fuzz_parity as fz
    seen = {}
    for line in open(os.path.join(ROOT, "tests", "tools", "fuzz_regressions.txt")):
        line = line.strip()
        if not line or line.startswith("#") or not ("'withNormalization': 0" in line or "'_dspace': True" in line):
            continue
        head, brace = line.split("{", 1)
        rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
        kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
        seen[(rows, cols, seed)] = fz.check(hip, orc, rows, cols, kw, scene, seed)
    print("\nun-normalised regression cases:", seen)
    with open(os.path.join(ROOT, "gpurun_out", "fuzz_outcomes.txt"), "a") as f:
        f.write("un-normalised regression cases: " + repr(seen) + "\n")
    assert len(seen) >= 12 and all(v in ACCEPTED for v in seen.values()), seen


def test_the_normalised_regression_cases_are_explained(hip, orc):
    """The two normalised cases the randomised runs of round 2 left unexplained (tests/tools/fuzz_regressions.txt), both stops of
    `|f - f_prev| < functionTolerance` on two IDENTICAL consecutive f32 values of f_norm:
     * 107x644 gradient / Tukey, 1.3e-2 rad from the oracle: at an iterate where the exact f changes by less than the rounding error of
       the sum — the oracle's f64 trace shows |f_k - f_(k-1)| <= 4e-6 f at that very iteration ("function-tol-at-the-noise-floor");
     * 187x206 descriptor fields, 1.14 x the bar: an iteration that wanders around its minimum in steps of 1e-3 of f until maxIterations
       on the oracle; the GPU's iterates are the oracle's to 9e-7 rad for 32 iterations, then its (correct: checked against exact sums
       at its own iterates) f_norm repeats and the reference's rule stops it there ("genuine-function-tol-stop").
     * (round 3) 97x133 descriptor fields / Tukey, 6e-4 rad after 50 iterations on both sides: at iteration 12 two consecutive robust
       scales of the GPU's run are 6e-7 apart and the scale freezes for the level (Q6), the oracle's, 3e-5 apart, does not; the GPU's
       two values are the oracle's own fresh estimates at the GPU's poses, bit for bit ("genuine-scale-freeze").
    In the first two the GPU's pose is the oracle's iterate of that moment."""
    import fuzz_parity as fz
    seen = {}
    for line in open(os.path.join(ROOT, "tests", "tools", "fuzz_regressions.txt")):
        line = line.strip()
        # (DisparitySpaceWarp::setNormalization is a no-op, bpvo/disparity_space_warp.h:87-90: a d-space case is an un-normalised one)
        if not line or line.startswith("#") or "'withNormalization': 1" not in line or "'_dspace': True" in line:
            continue
        head, brace = line.split("{", 1)
        rows, cols, scene, seed = (int(v) for v in head.split()[-4:])
        kw = ast.literal_eval("{" + brace.split("}", 1)[0] + "}")
        seen[(rows, cols)] = fz.check(hip, orc, rows, cols, kw, scene, seed)
    print("\nnormalised regression cases:", seen)
    assert seen and all(v in ACCEPTED for v in seen.values()), seen
    assert seen.get((107, 644)) in ("function-tol-at-the-noise-floor", "ok"), seen
    assert seen.get((187, 206)) in ("genuine-function-tol-stop", "ok"), seen
    assert seen.get((97, 133)) in ("genuine-scale-freeze", "ok"), seen
