// libbpvo_hip, host side: VisualOdometryPoseEstimator::estimatePose — the host drivers of the Gauss-Newton kernels (four-kernel chain with pipelined
// rounds, persistent single-pair kernel, team kernel), the operator-level seam (linearize, residuals, weights) and bpvo_hip_batch_estimate.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

namespace bpvo_hip_host {

// ---- estimatePose ---------------------------------------------------------------------------------------------------

// VisualOdometryPoseEstimator::estimatePose (reference: bpvo/vo_pose_estimator.cc:63-93) for a group of `n` workspaces on one
// lane.  wss[i]: workspace, refs[i] / curs[i]: frame slots.  T_init host [n][16] or null (Identity).
// Does a group of n pairs take the team-persistent kernel?  (kLinear, the f64 formulation, C = 8 or 1 like gn_persistent_kernel; not
// while per-kernel timings are being collected: there are no kernels to time)
// The shape of the team launch for a group of n pairs — ONE place decides it, for the launcher below and for team_serves: one workgroup per CU,
// teams of CUs / pairs workgroups (at most an eighth of the device's CUs, 32 of 256: teams of 64 were 8 - 12 % slower at 2 - 5 pairs, 4 % at 6 - 7; a
// team's workgroups stay dealt over ALL XCDs — pinned to one XCD each, 2 - 4 teams were 3 - 8 % slower), as many teams as fit; whether idle
// workgroups may join other teams (the growing form), and how many spare workgroups the division leaves for that.
struct TeamPlan { int team_size, n_teams, join_mode, spare_workgroups; };
static TeamPlan team_plan(const bpvo_hip_ctx* c, int n)
{
  TeamPlan t;
  const int slots = std::max(1, c->num_cus);
  int ts = c->team_size_env > 0 ? c->team_size_env : std::max(1, std::min(std::max(1, c->device_cus > 0 ? c->device_cus / 8 : 32), slots / std::max(1, n)));
  ts = std::max(1, std::min(ts, slots));
  t.team_size = ts;
  t.n_teams = std::max(1, std::min(std::min(n, slots / ts), kMaxTeams));
  // (nobody can join where the teams start at the admission cap or there is one team only: the leaders' looks at the tickets are then
  // skipped altogether — a 16-pair batch, teams of 16, lost 1.5 % to them)
  t.join_mode = (t.team_size < gn_team_max_size() && t.n_teams > 1 && n >= c->team_join_from_pairs) ? c->team_join : 0;
  // what the division CUs / pairs leaves over starts as spare workgroups that join the teams at their first admission
  t.spare_workgroups = (t.join_mode && c->team_spares) ? std::max(0, slots - t.team_size * t.n_teams) : 0;
  return t;
}

bool team_serves(const bpvo_hip_ctx* c, int n)
{
  bool size_ok = n >= 2 && n > c->persist_max_ws && n <= c->team_max_pairs;
  // above team_full_pairs only when the launch fills the chip: the kernel runs CUs / n workgroups per pair, and what that division leaves over
  // idles for the whole launch (96 pairs: 2 x 96 of 256 CUs, 770 k GN it/s against the chain's 840 k; 128 pairs: 2 x 128, 935 k against 890 k) —
  // unless the PLAN of this very launch turns the remainder into spare workgroups that join the teams (the growing form)
  if(size_ok && n > c->team_full_pairs && c->team_size_env <= 0 && c->num_cus > 0) {
    const TeamPlan t = team_plan(c, n);
    const int busy = t.team_size * t.n_teams + t.spare_workgroups;
    size_ok = 20 * busy >= 19 * c->num_cus;
  }
  return c->team_mode && c->persistent && !c->persistent_failed.load() && size_ok && !c->reference_reduction &&
         (c->C == 8 || c->C == 1) && c->params.interp == BPVO_INTERP_LINEAR && !c->fast_warp && !c->profile_all && !c->profile_k6_all &&
         c->num_cus >= 2 && g_live_ctx[c->device & 63].load() <= 1;
}

// ... and only templates the team kernel is good at: a pyramid level of more points than the persistent kernels are given (persist_max_points: dense
// templates, NMS off) is bandwidth work for the chain's chip-wide launches — the teams' per-pair median alone walks a thousand candidate segments
// per iteration there (640x480 bit-planes, NMS off, 4 / 16 / 64 pairs: 11.4 / 15.2 / 30.1 ms per step on the team kernel, 6.0 / 10.1 / 20.8 on the
// chain: scripts/dense_batch_ab.py)
static bool templates_suit_teams(const bpvo_hip_ctx* c, int n, const int* refs)
{
  for(int i = 0; i < n; ++i)
    for(int l = c->params.maxTestLevel; l < c->L; ++l)
      if(c->frames[refs[i]].n_host[l] > c->persist_max_points) return false;
  return true;
}

// allow_persistent: only a group that has the device to itself (a batch on ONE lane) may take the persistent kernel — two
// hand-barrier grids of concurrent lanes must not be co-scheduled.
int estimate_group(bpvo_hip_ctx* c, Lane* ln, int n, const int* wss, const int* refs, const int* curs, const float* T_init,
                   float* poses, bpvo_hip_stats* stats, float* d_records_out, bool allow_persistent)
{
  if(n <= 0) return BPVO_OK;
  (void) hipSetDevice(c->device);
  const bpvo_hip_params& p = c->params;
  const int NP = c->n_pairs;
  // (the pinned staging of a lane is free here: every call that uses it ends with a synchronisation of the lane's stream)
  std::vector<int> max_pts(c->L, 0);
  const size_t table = (size_t) c->L * NP;      // jobs of one table [L][NP]; wide descriptors: table 0 the whole jobs, tables 1 .. G their channel groups
  for(int l = 0; l < c->L; ++l)
    for(int i = 0; i < n; ++i) {
      PairJob& pj = ln->h_pjobs[(size_t) l * NP + i];
      pj = make_pair_job(c, wss[i], refs[i], curs[i], l);
      for(int k = 0; k < c->G && c->G > 1; ++k) ln->h_pjobs[(size_t) (1 + k) * table + (size_t) l * NP + i] = group_pair_job(c, pj, k);
      // Dense levels (no non-maximum suppression: most pixels are template points) gather their taps straight from the descriptor:
      // neighbouring points share three quarters of their footprints, so the 32-byte records are fetched about once per pixel
      // from HBM, where the per-point tap cache reads 128 bytes per point whatever the neighbours do.  The cache pays at the
      // sparse levels (one point in ~25 pixels: every footprint its own two or three lines).  Batches only: the persistent
      // single-pair kernel keeps its (L2-resident) cache.
      if((c->C == 8 || c->C == 1) && n > c->persist_max_ws && (double) pj.n > c->tapcache_max_density * (double) c->geom[l].npix) {
        pj.tapcache_on = 0;
      }
      max_pts[l] = std::max(max_pts[l], pj.n);
    }
  // validation mode "reference_reduction": the four-kernel chain with the index-order reduction in irls_reduce's place — residuals always written
  // (no fused path), the step in its own launch, no persistent / team kernel
  const bool ref_mode = c->reference_reduction != 0;
  const int fuse_frozen = ref_mode ? 0 : c->fuse_frozen;
  const bool pk_group = allow_persistent && c->persistent && !c->persistent_failed.load() && n <= c->persist_max_ws && !c->profile_all && !ref_mode;
  bool persistent = pk_group;
  // a context of a few pairs (one pair per call): table, poses and control words in one launch, the states copied out by the last one
  const bool small_ctx = (size_t) c->L * NP <= 64 && n <= 8 && c->small_batch_fused && c->G == 1;
  if(small_ctx) {
    if(T_init) std::memcpy(ln->h_T, T_init, sizeof(float) * 16 * n);
    launch_set_pose_upload(ln->stream, ln->d_pjobs, ln->h_pjobs, (size_t) c->L * NP, ln->h_pjobs + (size_t) (c->L - 1) * NP, T_init ? ln->h_T : nullptr, n,
                           persistent ? ln->d_pk_ctl : nullptr, persistent ? kPkCtlWords * kMaxLevels : 0);
  } else {
    const size_t table_bytes = sizeof(PairJob) * table * (size_t) job_tables(c);
    if(c->ctl_by_kernel.load()) launch_copy_rows(ln->stream, ln->d_pjobs, ln->h_pjobs, table_bytes, table_bytes, 1);
    else LANE_CK(ln, hipMemcpyAsync(ln->d_pjobs, ln->h_pjobs, table_bytes, hipMemcpyHostToDevice, ln->stream));
    const float* dT = nullptr;
    if(T_init) {
      std::memcpy(ln->h_T, T_init, sizeof(float) * 16 * n);
      if(c->ctl_by_kernel.load()) launch_copy_rows(ln->stream, ln->d_Tinit, ln->h_T, sizeof(float) * 16 * n, sizeof(float) * 16 * n, 1);
      else LANE_CK(ln, hipMemcpyAsync(ln->d_Tinit, ln->h_T, sizeof(float) * 16 * n, hipMemcpyHostToDevice, ln->stream));
      dT = ln->d_Tinit;
    }
    launch_set_pose(ln->stream, ln->d_pjobs + (size_t) (c->L - 1) * NP, dT, n, persistent ? ln->d_pk_ctl : nullptr, persistent ? kPkCtlWords * kMaxLevels : 0);
  }

  // PoseEstimatorParameters(AlgorithmParameters) (bpvo/pose_estimator_params.cc:27-33): maxFuncEvals stays 6*200 (Q4);
  // the low-res parameter set equals the full-res one (Q3).
  const int max_fun_evals = 6 * 200;
  // Small batches: the whole level loop in ONE launch, a team of workgroups per pair (gn_team_kernel)
  bool team_ran = false;
  // the normalisation of the levels below the coarsest, of the template stage queued just before (frames.hip, ctx->nrm_pending), may still be
  // running on the context's side stream: this lane waits for it where the second level starts — the team kernel, every level in one launch, at once
  // (level: the pyramid level about to start — the finest level's own event is joined only there; < 0: everything)
  auto join_normalization = [&](int level) -> int {
    if(c->nrm_pending) {
      LANE_CK(ln, hipStreamWaitEvent(ln->stream, c->nrm_pending, 0));
      c->nrm_pending = nullptr;
    }
    if(c->nrm_pending_finest && (level < 0 || level <= p.maxTestLevel)) {
      LANE_CK(ln, hipStreamWaitEvent(ln->stream, c->nrm_pending_finest, 0));
      c->nrm_pending_finest = nullptr;
    }
    return BPVO_OK;
  };
  bool team_split = false;
  if(allow_persistent && ln == &c->lanes[0] && team_serves(c, n) && templates_suit_teams(c, n, refs)) {
    // Small batches whose template stage left the normalisation of the levels below the coarsest on the side stream (frames.hip): the
    // team kernel in TWO launches — the coarsest level of every pair, then (behind that normalisation) the others.  The sums (0.17 - 0.19 ms
    // for a 1241x376 frame, whatever the batch) then run under the coarsest level's iterations instead of in front of the first one;
    // larger batches join them here, at once: a launch boundary makes every pair wait for the slowest (measured: 2 / 4 pairs + 1.2 / + 1.5 %,
    // 8 / 16 / 32 / 64 pairs - 5 / - 7 / - 4 / - 6 %: option "team_split_max_pairs", 4).
    team_split = (c->nrm_pending != nullptr || c->nrm_pending_finest != nullptr) && n <= c->team_split_max_pairs && c->L - 1 > p.maxTestLevel;
    if(!team_split)
      if(int rcj = join_normalization(-1)) return rcj;
    GNTeamLaunch t;
    t.jobs_all = ln->d_pjobs; t.job_pitch = NP; t.n_pairs = n; t.level_hi = c->L - 1; t.level_lo = p.maxTestLevel;
    t.C = c->C; t.loss = p.lossFunction; t.fuse_frozen = c->fuse_frozen;
    t.scale_is_moot = (p.lossFunction == BPVO_LOSS_L2 && c->C == 8 && c->fuse_frozen) ? 1 : 0;
    const TeamPlan plan = team_plan(c, n);      // (the same plan team_serves judged)
    t.team_size = plan.team_size;
    t.n_teams = plan.n_teams;
    t.ctl = ln->d_team_ctl;
    t.timeout_ticks = c->persist_timeout;
    t.local_barriers = c->team_local_barriers;
    t.join_mode = plan.join_mode;
    t.spare_workgroups = plan.spare_workgroups;
    LANE_CK(ln, hipMemsetAsync(ln->d_team_ctl, 0, sizeof(unsigned) * (size_t) gn_team_ctl_words(t.n_teams), ln->stream));
    hipError_t te;
    if(team_split) {
      t.level_lo = c->L - 1;
      te = launch_gn_team(ln->stream, t, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance);
      if(te == hipSuccess) {
        // (the first launch's control words: its abort word and join count are looked at with the second's)
        LANE_CK(ln, hipMemcpyAsync(ln->h_team_ctl + 32, ln->d_team_ctl, sizeof(unsigned) * 32, hipMemcpyDeviceToHost, ln->stream));
        launch_team_ctl_reset_keep_abort(ln->stream, ln->d_team_ctl, t.n_teams);      // (a first launch that gave up: the second leaves at once)
      }
      if(int rcj = join_normalization(-1)) return rcj;
      t.level_hi = c->L - 2; t.level_lo = p.maxTestLevel;
      if(te == hipSuccess) te = launch_gn_team(ln->stream, t, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance);
    } else {
      te = launch_gn_team(ln->stream, t, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance);
    }
    if(te == hipSuccess) {
      team_ran = true;
      c->team_launches.fetch_add(1);
      LANE_CK(ln, hipMemcpyAsync(ln->h_team_ctl, ln->d_team_ctl, sizeof(unsigned) * 32, hipMemcpyDeviceToHost, ln->stream));
    } else {
      (void) hipGetLastError();
      c->persistent_failed.store(true);      // degrade to the chain, now and for later calls
    }
  }
  auto level_launch = [&](int l) {
    GNLaunch g;
    g.jobs = ln->d_pjobs + (size_t) l * NP;
    g.npairs = n;
    g.max_points = max_pts[l];
    g.C = c->C;
    g.loss = p.lossFunction;
    g.fast_warp = c->fast_warp;
    g.interp = p.interp;
    g.fuse_frozen = fuse_frozen;
    g.step_in_reduce = (n <= c->step_in_reduce_max && !ref_mode && c->G == 1) ? 1 : 0;
    g.reference_reduction = ref_mode ? 1 : 0;
    g.dense_candidates = dense_candidates(c, g.max_points);
    g.step_prm = GNParams{p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance};
    return g;
  };
  // kL2: the weights are 1 whatever the robust scale — with the fused path every linearisation is irls_reduce + gn_step only
  const bool l2_moot = p.lossFunction == BPVO_LOSS_L2 && c->C == 8 && fuse_frozen && !c->fast_warp && p.interp == BPVO_INTERP_LINEAR;
  bool begun = false;      // this level's start was left to its persistent kernel (its tap-cache keys invalidated by the kernel of the level before)
  for(int l = c->L - 1; l >= p.maxTestLevel && !team_ran; --l) {
    GNLaunch g = level_launch(l);
    if(l < c->L - 1)
      if(int rcj = join_normalization(l)) return rcj;
    const bool was_begun = begun;
    begun = false;
    if(!was_begun) launch_level_begin(ln->stream, g.jobs, n, g.max_points, l, l2_moot ? 1 : 0);    // (and the tap-cache keys of the level)
    if(g.max_points <= 0) continue;
    if(persistent && gn_persistent_serves(g) && g.max_points <= c->persist_max_points) {
      // the whole level in one launch — and the start of the next level with it, when that one takes the kernel too
      const bool next_too = l - 1 >= p.maxTestLevel && max_pts[l - 1] > 0 && gn_persistent_serves(level_launch(l - 1)) && max_pts[l - 1] <= c->persist_max_points;
      g.begin_level = was_begun ? l : -1;
      g.begin_moot = l2_moot ? 1 : 0;
      g.next_jobs = next_too ? ln->d_pjobs + (size_t) (l - 1) * NP : nullptr;
      const hipError_t pe = launch_gn_persistent(ln->stream, g, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance,
                                                 ln->d_pk_ctl + (size_t) l * kPkCtlWords, gn_persistent_grid(g, c->persist_grid), c->persist_timeout);
      if(pe == hipSuccess) {
        c->persistent_levels.fetch_add(1);
        begun = next_too;
        continue;
      }
      // the device cannot grant the kernel its LDS / residency (or the launch failed): degrade to the four-kernel chain — this level,
      // the rest of the pyramid and every later call of the context — instead of failing the estimate
      (void) hipGetLastError();
      c->persistent_failed.store(true);
      if(was_begun) launch_level_begin(ln->stream, g.jobs, n, g.max_points, l, l2_moot ? 1 : 0);
    }
    persistent = false;     // (a level the kernel does not serve: the rest of the pyramid takes the chain as well)
    // At most maxIterations + 2 linearisations per level (pose_estimator_base.h:373-393); the state machine on the device
    // enforces the limits, the host queues rounds of kItersPerSync iterations until the device reports no active workspace.
    // Every round ends with a compaction of the list of still-active workspaces (ActiveSet, kernels.h) and the copy of its
    // count.  The rounds are PIPELINED: round r + 1 is queued with the list and count that came out of round r - 1, as soon
    // as those have landed — the device never waits for the host (a synchronisation per round was a ~30 us bubble: 7 % of a
    // round at 128 pairs, 11 % for a single pair).  Workspaces that finished in between are still dispatched for one more
    // round (their workgroups exit on the first load), and the level ends with one round of empty launches.
    const int max_lin = std::min(p.maxIterations + 2, max_fun_evals);
    const int kItersPerSync = 4;
    const int max_rounds = (max_lin + kItersPerSync - 1) / kItersPerSync + 2;
    constexpr unsigned kProfileEvery = 5;   // co-prime with kItersPerSync: no phase lock with the host round trips
    int* const lists[3] = {ln->d_list, ln->d_list + NP, ln->d_list + 2 * (size_t) NP};
    int n_cur = n;
    g.active.list = nullptr;                // first rounds: every workspace of the group, in order
    // Once NO active workspace of the list estimates its robust scale any more (a frozen scale stays frozen for the level, and the
    // list only shrinks), the median has nothing to do and — with the fused path, where irls_reduce recomputes the residuals of
    // frozen workspaces itself — neither has warp_residual: their launches are dropped for the rest of the level.  (Each would
    // still cost its floor of ~5 us per iteration in the tail of a level.)  l2_moot: true from the first linearisation.
    const bool fused_path = c->C == 8 && fuse_frozen && !c->fast_warp && p.interp == BPVO_INTERP_LINEAR;
    bool none_moving = l2_moot;
    for(int round = 0; round < max_rounds; ++round) {
      g.npairs = n_cur;
      const bool launch_median_k = !none_moving, launch_warp_k = !(none_moving && fused_path);
      for(int k = 0; k < kItersPerSync; ++k) {
        // level 1 brackets every kProfileEvery-th warp_residual launch of the lane with events (a running counter, so the
        // sampled launches rotate through all iterations and levels): an event pair costs a few µs of dispatch gap
        if(launch_warp_k) {
          const bool sampled = c->profile_all || c->profile_k6_all || (ln->k6_seq++ % kProfileEvery) == 0;
          ScopedTimer t(c, KC_WARP_RESIDUAL, 0.0, ln, sampled);
          for_each_group(c, g, table, [&](const GNLaunch& gg) { launch_warp_residual(ln->stream, gg); });
        }
        // level 2 times every kernel; level 3 (bench.py's single-lane roofline pass) every warp_residual AND every irls_reduce launch
        { ScopedTimer t(c, KC_MEDIAN, 0.0, ln, c->profile_all && launch_median_k); if(launch_median_k) launch_median(ln->stream, median_launch(c, g)); }
        {
          ScopedTimer t(c, KC_IRLS_REDUCE, 0.0, ln, c->profile_all || c->profile_k6_all);
          if(ref_mode) launch_irls_reduce(ln->stream, g);      // (reference order: one wave per pair walks every channel of the whole job)
          else for_each_group(c, g, table, [&](const GNLaunch& gg) { launch_irls_reduce(ln->stream, gg); });
        }
        if(!g.step_in_reduce) {
          ScopedTimer t(c, KC_GN_STEP, 0.0, ln, c->profile_all);
          launch_gn_step(ln->stream, g, 0, p.maxIterations, max_fun_evals, p.parameterTolerance, p.functionTolerance, p.gradientTolerance);
        }
      }
      const int slot = round % 3;
      launch_compact_active(ln->stream, g.jobs, g.active, n_cur, lists[slot], ln->d_active + 2 * slot);
      LANE_CK(ln, hipMemcpyAsync(ln->h_active + 2 * slot, ln->d_active + 2 * slot, 2 * sizeof(int), hipMemcpyDeviceToHost, ln->stream));
      LANE_CK(ln, hipEventRecord(ln->round_ev[slot], ln->stream));
      if(round == 0) continue;              // nothing to learn yet: queue the second round behind the first
      const int prev = (round - 1) % 3;
      LANE_CK(ln, hipEventSynchronize(ln->round_ev[prev]));
      const int n_prev = ln->h_active[2 * prev];
      if(n_prev <= 0) break;                // (the round just queued runs empty)
      n_cur = n_prev;
      none_moving = none_moving || ln->h_active[2 * prev + 1] == 0;
      g.active.list = lists[prev];
    }
  }
  const bool want_frac = n == 1 && c->prefetch_frac_thr >= 0.0f && ln == &c->lanes[0];
  if(small_ctx) launch_pack_records(ln->stream, ln->d_pjobs + (size_t) (c->L - 1) * NP, n, c->L, d_records_out, c->d_states, ln->h_states, pk_group ? ln->d_pk_ctl : nullptr,
                                    pk_group ? ln->h_pk_ctl : nullptr, kPkCtlWords * kMaxLevels, want_frac ? c->d_count : nullptr);
  else launch_pack_records(ln->stream, ln->d_pjobs + (size_t) (c->L - 1) * NP, n, c->L, d_records_out);
  bool frac_queued = false;
  if(want_frac) {
    // fraction_good of this workspace at the level the estimate ended on, from the job already on the device
    const PairJob* job = ln->d_pjobs + (size_t) p.maxTestLevel * NP;
    const int npts = ln->h_pjobs[(size_t) p.maxTestLevel * NP].n;
    if(npts > 0) {
      GNLaunch gr;
      gr.jobs = job; gr.npairs = 1; gr.max_points = npts; gr.C = c->C;
      launch_refresh_residuals(ln->stream, gr);
      if(!small_ctx) LANE_CK(ln, hipMemsetAsync(c->d_count, 0, sizeof(unsigned int), ln->stream));      // (small contexts: cleared by pack_records)
      launch_count_good(ln->stream, job, npts, c->C, p.lossFunction, c->prefetch_frac_thr, c->d_count);
      LANE_CK(ln, hipMemcpyAsync(c->h_ints, c->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, ln->stream));
      frac_queued = true;
      c->frac_n = npts;
    }
  }
  // only this group's states: other lanes may still be writing theirs
  int ws_lo = wss[0], ws_hi = wss[0];
  for(int i = 1; i < n; ++i) { ws_lo = std::min(ws_lo, wss[i]); ws_hi = std::max(ws_hi, wss[i]); }
  if(!small_ctx) {
    LANE_CK(ln, hipMemcpyAsync(ln->h_states + ws_lo, c->d_states + ws_lo, sizeof(GNState) * (size_t) (ws_hi - ws_lo + 1), hipMemcpyDeviceToHost, ln->stream));
    if(pk_group)
      LANE_CK(ln, hipMemcpyAsync(ln->h_pk_ctl, ln->d_pk_ctl, sizeof(unsigned) * kPkCtlWords * kMaxLevels, hipMemcpyDeviceToHost, ln->stream));
  }
  if(c->before_final_sync && ln == &c->lanes[0]) {      // (every kernel of the estimate is queued: the host is free for a moment)
    const std::function<int()> f = std::move(c->before_final_sync);
    c->before_final_sync = nullptr;
    if(const int rcf = f()) return rcf;
  }
  LANE_CK(ln, hipStreamSynchronize(ln->stream));
  LANE_CK(ln, hipGetLastError());
  if(frac_queued) { c->frac_valid = true; c->frac_ws = wss[0]; c->frac_thr = c->prefetch_frac_thr; c->frac_cnt = (unsigned) c->h_ints[0]; }
#ifdef BPVO_PK_TIMING
  if(pk_group) {     // library built with -DBPVO_PK_TIMING: per-phase ticks (10 ns) of workgroup 0
    static const char* names[6] = {"warp", "barrier1", "median", "irls", "barrier2", "step"};
    for(int l = c->L - 1; l >= 0; --l) {
      const unsigned* t = ln->h_pk_ctl + (size_t) l * kPkCtlWords;
      if(!t[15]) continue;
      std::fprintf(stderr, "pk level %d: %u iterations;", l, t[15]);
      for(int k = 0; k < 6; ++k) std::fprintf(stderr, " %s %.2f", names[k], 0.01 * t[8 + k] / t[15]);
      std::fprintf(stderr, " | step: sum_partials %.2f unpack %.2f solve %.2f pose %.2f tests %.2f", 0.01 * t[20] / t[15], 0.01 * t[16] / t[15], 0.01 * t[17] / t[15],
                   0.01 * t[18] / t[15], 0.01 * t[19] / t[15]);
      std::fprintf(stderr, " us per iteration\n");
    }
  }
  if(team_ran) {       // phases of team 0's first workgroup
    static const char* names[6] = {"warp", "barrier1", "median", "irls", "barrier2", "step"};
    for(int l = c->L - 1; l >= 0 && l < 4; --l) {
      const unsigned* t = ln->h_team_ctl + 4 + 7 * l;
      if(!t[6]) continue;
      std::fprintf(stderr, "team level %d: %u iterations;", l, t[6]);
      for(int k = 0; k < 6; ++k) std::fprintf(stderr, " %s %.2f", names[k], 0.01 * t[k] / t[6]);
      std::fprintf(stderr, " us per iteration\n");
    }
  }
#endif
  if(team_ran) c->team_joins.fetch_add(ln->h_team_ctl[3] + (team_split ? ln->h_team_ctl[32 + 3] : 0u));      // workgroups that joined another team (measurement)
  if(team_ran && (ln->h_team_ctl[1] != 0 || (team_split && ln->h_team_ctl[32 + 1] != 0))) {
    // a team barrier timed out (teams not co-resident): rerun the group through the four-kernel chain and stay on it
    c->persistent_failed.store(true);
    return estimate_group(c, ln, n, wss, refs, curs, T_init, poses, stats, d_records_out, false);
  }
  if(pk_group) {
    bool gave_up = false;
    for(int l = 0; l < c->L; ++l) gave_up = gave_up || ln->h_pk_ctl[(size_t) l * kPkCtlWords + 1] != 0;
    if(gave_up) {
      // a barrier timed out: the states of that level were not written back.  Rerun the group through the four-kernel chain
      // (same results) and keep this context on it.
      c->persistent_failed.store(true);
      return estimate_group(c, ln, n, wss, refs, curs, T_init, poses, stats, d_records_out, false);
    }
  }
  for(int i = 0; i < n; ++i) {
    const GNState& st = ln->h_states[wss[i]];
    if(poses) std::memcpy(poses + 16 * (size_t) i, st.T_out, 16 * sizeof(float));
    if(stats)
      for(int l = 0; l < c->L; ++l) stats[(size_t) i * c->L + l] = st.stats[l];
  }
  return BPVO_OK;
}

int estimate_batch(bpvo_hip_ctx* c, int n, const int* wss, const int* refs, const int* curs, const float* T_init, float* poses,
                   bpvo_hip_stats* stats)
{
  if(n <= 0) return BPVO_OK;
  if(n > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "more pairs than workspaces");
  const bpvo_hip_params& p = c->params;
  for(int i = 0; i < n; ++i) {
    if(wss[i] < 0 || wss[i] >= c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad workspace");
    if(refs[i] < 0 || refs[i] >= c->n_frames || curs[i] < 0 || curs[i] >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot");
    if(!c->frames[refs[i]].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "reference frame has no template");
    if(!c->frames[curs[i]].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
    if(int rc = ensure_dense_descriptor(c, curs[i])) return rc;      // (a batch's template frame as the current frame of this estimate)
  }
  int nl = lanes_for(c, n, 8);
  if(nl < 0) return nl;
  if(team_serves(c, n) && templates_suit_teams(c, n, refs)) nl = 1;      // the team-persistent kernel takes the whole chip
  // frame stages run on the ctx stream: the other lanes' streams start from a quiet device.  A single lane IS the ctx stream — its
  // launches simply queue behind the frame stage (sequential addFrame: ~30 us of idle device per frame otherwise).
  if(nl > 1) {
    HIP_CK(c, join_pending_normalization(c, c->stream));
    HIP_CK(c, hipStreamSynchronize(c->stream));
  }
  std::vector<int> rcs(nl, BPVO_OK);
  c->frac_valid = false;      // (on the API thread: the lane threads only read the context's settings)
  auto run = [&](int k) {
    const int lo = (int) ((long long) n * k / nl), hi = (int) ((long long) n * (k + 1) / nl);
    rcs[k] = estimate_group(c, &c->lanes[k], hi - lo, wss + lo, refs + lo, curs + lo, T_init ? T_init + 16 * (size_t) lo : nullptr,
                            poses ? poses + 16 * (size_t) lo : nullptr, stats ? stats + (size_t) lo * c->L : nullptr,
                            c->d_records + (size_t) kRecordFloats * lo, nl == 1);
  };
  if(nl == 1) {
    run(0);
  } else {
    std::vector<std::thread> th;
    for(int k = 1; k < nl; ++k) th.emplace_back(run, k);
    run(0);
    for(auto& t : th) t.join();
  }
  for(int k = 0; k < nl; ++k)
    if(rcs[k]) { c->err = c->lanes[k].err; return rcs[k]; }
  resolve_events(c);
  for(int i = 0; i < n; ++i) {
    Workspace& w = c->ws[wss[i]];
    w.last_ref = refs[i];
    w.last_cur = curs[i];
    w.last_level = p.maxTestLevel;
  }
  return BPVO_OK;
}

// tiled device layout (types.h tile_index) -> reference channel-major layout: out[(ch*n + i)*E + e] for records of
// C*E floats per point cut in V-float pieces
void detile_to_channel_major(const float* src, int n, int C, int E, int V, float* out)
{
  const int W = C * E, pieces = W / V;
  for(int i = 0; i < n; ++i)
    for(int w = 0; w < W; ++w) {
      const int piece = w / V, within = w - piece * V;
      const float v = src[(((size_t) (i / kTile) * pieces + piece) * kTile + (size_t) (i % kTile)) * V + within];
      const int ch = w / E, e = w - ch * E;
      out[((size_t) ch * n + i) * E + e] = v;
    }
}
size_t tiled_floats(int n, int floats_per_point) { return (size_t) ((n + kTile - 1) / kTile) * kTile * floats_per_point; }

int refresh_counters(bpvo_hip_ctx* c)
{
  // per-workspace counters (PairJob::cnt), summed here
  std::vector<unsigned long long> all(kWsCounters * (size_t) c->n_pairs);
  HIP_CK(c, hipMemcpy(all.data(), c->d_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  unsigned long long h[kWsCounters] = {};
  for(int w = 0; w < c->n_pairs; ++w)
    for(int k = 0; k < kWsCounters; ++k) h[k] += all[kWsCounters * (size_t) w + k];
  c->median_bracketed = h[2];
  c->median_full = h[3];
  for(int k = 0; k < 4; ++k) c->tap_counts[k] = h[5 + k];
  c->total_lin = h[1];
  // units of the GN kernels = points linearised (device-side count: only the pairs still active in a launch count)
  // warp_residual at profiling level 1 is timed on a 1-in-kProfileEvery sample of its launches: its units are scaled to
  // the sampled launches so that units / launches stays the points of an average launch
  double all_k6 = 0;
  for(const auto& ln : c->lanes) all_k6 += ln.k6_seq;
  const bool sampled = c->profiling && !c->profile_all && !c->profile_k6_all && all_k6 > 0;
  // (h[4]: the points warp_residual itself processed; workspaces with a frozen scale go through irls_reduce's fused path)
  c->kc_units[KC_WARP_RESIDUAL] = sampled ? (double) h[4] * (double) c->kc_launches[KC_WARP_RESIDUAL] / all_k6 : (double) h[4];
  c->points_fused = (double) h[10];
  c->kc_units[KC_IRLS_REDUCE] = (double) h[0];
  c->kc_units[KC_MEDIAN] = (double) h[0];
  c->kc_units[KC_GN_STEP] = (double) h[1];
  return BPVO_OK;
}

int upload_single_job(bpvo_hip_ctx* c, int ws, int ref, int cur, int level)
{
  PairJob* h = c->lanes[0].h_pjobs;
  h[0] = make_pair_job(c, ws, ref, cur, level);
  for(int k = 0; k < c->G && c->G > 1; ++k) h[1 + k] = group_pair_job(c, h[0], k);      // (wide descriptors: d_job1[1 + k] = channel group k)
  HIP_CK(c, hipMemcpyAsync(c->d_job1, h, sizeof(PairJob) * (size_t) job_tables(c), hipMemcpyHostToDevice, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));   // h_pjobs is reused by the next call
  return BPVO_OK;
}

void trajectory_push(bpvo_hip_ctx* c, const M44& T)   // Trajectory::push_back + InvertPose (bpvo/trajectory.cc:30-50)
{
  M44 Ti = m44_identity();
  for(int i = 0; i < 3; ++i)
    for(int j = 0; j < 3; ++j) Ti.m[i * 4 + j] = T.m[j * 4 + i];
  for(int i = 0; i < 3; ++i) {
    float s = Ti.m[0 * 4 + i] * T.m[3];
    s += Ti.m[1 * 4 + i] * T.m[7];
    s += Ti.m[2 * 4 + i] * T.m[11];
    Ti.m[i * 4 + 3] = -s;
  }
  if(!c->trajectory.empty()) c->trajectory.push_back(m44_mul(c->trajectory.back(), Ti));
  else c->trajectory.push_back(Ti);
}

// Fused path of the estimate loops: the residual / valid buffers of a workspace may lag behind its last linearisation
// (GNState::r_stale).  Everything that reads them goes through here first; the check itself happens on the device.
int ensure_residuals(bpvo_hip_ctx* c, int ws)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0 || c->C != 8) return BPVO_OK;
  int rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  GNLaunch g;
  g.jobs = c->d_job1; g.npairs = 1; g.max_points = c->frames[w.last_ref].n_host[w.last_level]; g.C = c->C;
  launch_refresh_residuals(c->stream, g);
  return BPVO_OK;
}

int fraction_good(bpvo_hip_ctx* c, int ws, float thr, float* frac)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  if(c->frac_valid && c->frac_ws == ws && c->frac_thr == thr && c->frac_n == n) {      // queued behind the estimate by addFrame
    *frac = c->frac_cnt / static_cast<float>((size_t) n * c->C);
    return BPVO_OK;
  }
  int rc = ensure_residuals(c, ws);
  if(rc) return rc;
  rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  HIP_CK(c, hipMemsetAsync(c->d_count, 0, sizeof(unsigned int), c->stream));
  launch_count_good(c->stream, c->d_job1, n, c->C, c->params.lossFunction, thr, c->d_count);
  unsigned int cnt = 0;
  HIP_CK(c, hipMemcpyAsync(c->h_ints, c->d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  cnt = (unsigned int) c->h_ints[0];
  *frac = cnt / static_cast<float>((size_t) n * c->C);   // vo_pose_estimator.cc:105-106
  return BPVO_OK;
}

int get_weights_host(bpvo_hip_ctx* c, int ws, std::vector<float>& w_cm, int* n_out)
{
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  const int C = c->C;
  int rc = ensure_residuals(c, ws);
  if(rc) return rc;
  rc = upload_single_job(c, ws, w.last_ref, w.last_cur, w.last_level);
  if(rc) return rc;
  launch_weights(c->stream, c->d_job1, n, C, c->params.lossFunction, c->d_wtmp);
  std::vector<float> pm((size_t) n * C);
  HIP_CK(c, hipMemcpyAsync(pm.data(), c->d_wtmp, pm.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  w_cm.resize(pm.size());
  for(int i = 0; i < n; ++i)
    for(int ch = 0; ch < C; ++ch) w_cm[(size_t) ch * n + i] = pm[(size_t) i * C + ch];
  *n_out = n;
  return BPVO_OK;
}

int check_template_not_empty(bpvo_hip_ctx* c, int ref_slot)
{
  if(ref_slot < 0 || ref_slot >= c->n_frames || !c->frames[ref_slot].has_template) return BPVO_OK;   // reported elsewhere
  for(int l = c->params.maxTestLevel; l < c->L; ++l)
    if(c->frames[ref_slot].n_host[l] <= 0) return fail(c, BPVO_ERR_NO_TEMPLATE, "you should call setData before calling computeResiduals");
  return BPVO_OK;
}

}  // namespace bpvo_hip_host

extern "C" {

// ---- operator-level seam --------------------------------------------------------------------------------------------
static int linearize_impl(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], int reset_scale, float given_scale,
                          float H[36], float G[6], float* f_norm, float* sigma, int* num_valid)
{
  CHECK_CTX(c); CHECK_WS(c, ws); CHECK_SLOT(c, ref_slot); CHECK_SLOT(c, cur_slot); CHECK_LEVEL(c, level);
  if(!c->frames[ref_slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "reference frame has no template");
  if(!c->frames[cur_slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  if(int rc0 = ensure_dense_descriptor(c, cur_slot)) return rc0;
  if(c->frames[ref_slot].n_host[level] <= 0) return fail(c, BPVO_ERR_NO_TEMPLATE, "you should call setData before calling computeResiduals");
  c->frac_valid = false;
  (void) hipSetDevice(c->device);
  int rc = upload_single_job(c, ws, ref_slot, cur_slot, level);
  if(rc) return rc;
  Lane& l0 = c->lanes[0];
  std::memcpy(l0.h_T, T, 16 * sizeof(float));
  HIP_CK(c, hipMemcpyAsync(l0.d_Tinit, l0.h_T, 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  launch_prepare_linearize(c->stream, c->d_job1, l0.d_Tinit, reset_scale, level, given_scale);
  GNLaunch g;
  g.jobs = c->d_job1; g.npairs = 1; g.max_points = c->frames[ref_slot].n_host[level]; g.C = c->C; g.loss = c->params.lossFunction;
  g.fast_warp = c->fast_warp;
  g.interp = c->params.interp;
  g.reference_reduction = c->reference_reduction ? 1 : 0;
  g.dense_candidates = dense_candidates(c, g.max_points);
  launch_reset_tapkeys(c->stream, g);
  { ScopedTimer t(c, KC_WARP_RESIDUAL, 0.0); for_each_group(c, g, 1, [&](const GNLaunch& gg) { launch_warp_residual(c->stream, gg); }); }
  { ScopedTimer t(c, KC_MEDIAN, 0.0); launch_median(c->stream, median_launch(c, g)); }
  {
    ScopedTimer t(c, KC_IRLS_REDUCE, 0.0);
    if(g.reference_reduction) launch_irls_reduce(c->stream, g);
    else for_each_group(c, g, 1, [&](const GNLaunch& gg) { launch_irls_reduce(c->stream, gg); });
  }
  { ScopedTimer t(c, KC_GN_STEP, 0.0); launch_gn_step(c->stream, g, 1, 0, 0, 0, 0, 0); }
  HIP_CK(c, hipMemcpyAsync(l0.h_states, c->d_states + ws, sizeof(GNState), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  HIP_CK(c, hipGetLastError());
  resolve_events(c);
  const GNState& st = l0.h_states[0];
  std::memcpy(H, st.H, sizeof(st.H));
  std::memcpy(G, st.G, sizeof(st.G));
  *f_norm = st.f_norm;
  if(sigma) *sigma = st.scale;
  *num_valid = (int) st.n_valid;
  c->ws[ws].last_ref = ref_slot; c->ws[ws].last_cur = cur_slot; c->ws[ws].last_level = level;
  return BPVO_OK;
}
int bpvo_hip_linearize(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], int reset_scale,
                       float H[36], float G[6], float* f_norm, float* sigma, int* num_valid)
{
  return linearize_impl(c, ws, ref_slot, cur_slot, level, T, reset_scale ? 1 : 0, 0.0f, H, G, f_norm, sigma, num_valid);
}
int bpvo_hip_linearize_at_scale(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, int level, const float T[16], float sigma,
                                float H[36], float G[6], float* f_norm, int* num_valid)
{
  if(c && !(sigma > 0.0f)) return fail(c, BPVO_ERR_INVALID_ARG, "sigma must be positive");
  return linearize_impl(c, ws, ref_slot, cur_slot, level, T, 2, sigma, H, G, f_norm, nullptr, num_valid);
}

int bpvo_hip_get_residuals(bpvo_hip_ctx* c, int ws, float* r, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level], C = c->C;
  if(n_out) *n_out = (size_t) n * C;
  if(!r) return BPVO_OK;
  (void) hipSetDevice(c->device);
  { int rc = ensure_residuals(c, ws); if(rc) return rc; }
  std::vector<float> t(tiled_floats(n, C));
  if(n) HIP_CK(c, hipMemcpyAsync(t.data(), w.r, t.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  detile_to_channel_major(t.data(), n, C, 1, C == 8 ? 4 : C, r);
  return BPVO_OK;
}
int bpvo_hip_get_valid(bpvo_hip_ctx* c, int ws, uint16_t* v, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& w = c->ws[ws];
  if(w.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n = c->frames[w.last_ref].n_host[w.last_level];
  if(n_out) *n_out = (size_t) n;
  if(!v) return BPVO_OK;
  (void) hipSetDevice(c->device);
  { int rc = ensure_residuals(c, ws); if(rc) return rc; }
  std::vector<uint8_t> b((size_t) n);
  HIP_CK(c, hipMemcpyAsync(b.data(), w.valid, b.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  for(int i = 0; i < n; ++i) v[i] = b[i];
  return BPVO_OK;
}
int bpvo_hip_get_weights(bpvo_hip_ctx* c, int ws, float* w, size_t* n_out)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  Workspace& wk = c->ws[ws];
  if(wk.last_ref < 0) return fail(c, BPVO_ERR_NO_DATA, "no linearisation has run on this workspace");
  const int n0 = c->frames[wk.last_ref].n_host[wk.last_level];
  if(n_out) *n_out = (size_t) n0 * c->C;
  if(!w) return BPVO_OK;
  (void) hipSetDevice(c->device);
  std::vector<float> cm;
  int n = 0;
  int rc = get_weights_host(c, ws, cm, &n);
  if(rc) return rc;
  std::memcpy(w, cm.data(), cm.size() * sizeof(float));
  return BPVO_OK;
}
int bpvo_hip_fraction_good(bpvo_hip_ctx* c, int ws, float threshold, float* frac)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  (void) hipSetDevice(c->device);
  return fraction_good(c, ws, threshold, frac);
}

// TemplateData::computeResiduals throws on an empty template (reference: bpvo/template_data.cc:177).  The single-pair entry
// points mirror that; the batch entry points skip such levels of the affected pair (its statistics keep kSolverError).

int bpvo_hip_set_warp_formulation(bpvo_hip_ctx* c, int mode)
{
  CHECK_CTX(c);
  if(mode != BPVO_WARP_PHOTO_ERROR_F64 && mode != BPVO_WARP_PROJECT_POINTS_F32 && mode != BPVO_WARP_DISPARITY_SPACE_F32)
    return fail(c, BPVO_ERR_INVALID_ARG, "unknown warp formulation");
  if(mode != BPVO_WARP_PHOTO_ERROR_F64 && c->params.interp != BPVO_INTERP_LINEAR)
    return fail(c, BPVO_ERR_UNSUPPORTED, "the f32 formulations are kLinear only (bpvo/photo_error.cc:118-214)");
  const int dspace = (mode == BPVO_WARP_DISPARITY_SPACE_F32) ? 1 : 0;
  if(dspace != c->dspace) {
    // templates hold the points / gradients of the other warp: they have to be rebuilt (frame data stays)
    for(auto& f : c->frames) f.has_template = false;
  }
  c->dspace = dspace;
  c->fast_warp = (mode != BPVO_WARP_PHOTO_ERROR_F64) ? 1 : 0;
  return BPVO_OK;
}

int bpvo_hip_estimate_pose(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                           bpvo_hip_stats* stats)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  if(!T_init || !T_est) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr pose");
  if(int rc = check_template_not_empty(c, ref_slot)) return rc;
  (void) hipSetDevice(c->device);
  return estimate_batch(c, 1, &ws, &ref_slot, &cur_slot, T_init, T_est, stats);
}

int bpvo_hip_estimate_pose_trace(bpvo_hip_ctx* c, int ws, int ref_slot, int cur_slot, const float T_init[16], float T_est[16],
                                 bpvo_hip_stats* stats, float* records, int max_records, int* n_records)
{
  CHECK_CTX(c); CHECK_WS(c, ws);
  if(!T_init || !T_est || !n_records || max_records < 0 || (max_records > 0 && !records)) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr pose / records");
  if(int rc = check_template_not_empty(c, ref_slot)) return rc;
  (void) hipSetDevice(c->device);
  // at most min(maxIterations + 2, maxFuncEvals) linearisations per level (pose_estimator_base.h:373-393)
  const int cap = c->L * (std::min(std::max(c->params.maxIterations, 0) + 2, 6 * 200) + 1);
  if(cap > c->trace_cap) {
    HIP_CK(c, hipStreamSynchronize(c->stream));
    (void) hipFree(c->d_trace);
    c->d_trace = nullptr; c->trace_cap = 0;
    HIP_CK(c, hipMalloc((void**) &c->d_trace, sizeof(float) * kTraceFloats * (size_t) cap));
    c->trace_cap = cap;
  }
  c->trace_ws = ws;
  const int rc = estimate_batch(c, 1, &ws, &ref_slot, &cur_slot, T_init, T_est, stats);
  c->trace_ws = -1;
  if(rc) return rc;
  const int n = c->lanes[0].h_states[ws].trace_n;
  *n_records = n;
  const int ncopy = std::min(std::min(n, max_records), c->trace_cap);
  if(ncopy > 0) HIP_CK(c, hipMemcpy(records, c->d_trace, sizeof(float) * kTraceFloats * (size_t) ncopy, hipMemcpyDeviceToHost));
  return BPVO_OK;
}

// ---- batches --------------------------------------------------------------------------------------------------------
int bpvo_hip_batch_estimate(bpvo_hip_ctx* c, int n_pairs, const float* T_init, float* poses, bpvo_hip_stats* stats)
{
  CHECK_CTX(c);
  if(n_pairs < 0 || 2 * n_pairs > c->n_frames || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "batch exceeds ctx capacity");
  (void) hipSetDevice(c->device);
  std::vector<int> wss(n_pairs), refs(n_pairs), curs(n_pairs);
  for(int p = 0; p < n_pairs; ++p) { wss[p] = p; refs[p] = 2 * p; curs[p] = 2 * p + 1; }
  return estimate_batch(c, n_pairs, wss.data(), refs.data(), curs.data(), T_init, poses, stats);
}

}  // extern "C"
