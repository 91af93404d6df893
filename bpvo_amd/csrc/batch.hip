// libbpvo_hip, host side: pair batches — bpvo_hip_batch_run on staggered lanes, the upload pipeline of host-buffer batches, the result records.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

namespace {

// ---- upload pipeline of host-buffer batches ---------------------------------------------------------------------------------
// The caller's buffers are pageable: a hipMemcpyAsync from them is staged by the runtime through ONE thread's memcpy (a few GB/s) on the
// stream that should be computing.  Here up_workers threads copy chunks of kUploadChunkPairs pairs (both images, the disparity of the
// template frame A only) into pinned slots of their own and hand them to the copy engines on their own streams; a lane's frame stage
// takes the chunks of its pairs as they land (one stream-wait per chunk) and ingests them from the device staging area.  The first chunk
// is all the device ever waits for; the rest of the upload runs under the compute of the chunks before it.
constexpr int kUploadChunkPairs = 16;
constexpr int kUploadGroup = 4;      // chunks a lane's frame stage takes at once
struct UploadRun {
  bpvo_hip_ctx* c = nullptr;
  int n_pairs = 0;
  std::vector<std::pair<int, int>> chunks;     // [first pair, count), lane after lane
  std::vector<int> recorded;                   // 1: the chunk's event has been recorded (or the worker failed: error set)
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> workers;
  std::string err;
  ~UploadRun() { for(auto& t : workers) if(t.joinable()) t.join(); }
};

int upload_prepare(bpvo_hip_ctx* c, int n_pairs)
{
  const size_t npix = c->geom[0].npix;
  // one worker = one pinned block of two slots + the events that free them (the copy stream is shared: up_streams[0]).  A worker's
  // resources are taken into the context only when all of them exist: a failed pinned allocation leaves the vectors in step
  c->up_slot_bytes = (size_t) kUploadChunkPairs * npix * (2 + 4);
  if(c->up_streams.empty()) {
    hipStream_t st = nullptr;
    HIP_CK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    c->up_streams.push_back(st);
  }
  while((int) c->up_pinned.size() < c->up_workers) {
    uint8_t* pin = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t e = hipHostMalloc((void**) &pin, 2 * c->up_slot_bytes);
    for(int sl = 0; sl < 2 && e == hipSuccess; ++sl) e = hipEventCreateWithFlags(&ev[sl], hipEventDisableTiming);
    if(e != hipSuccess) {
      if(pin) (void) hipHostFree(pin);
      for(auto x : ev) if(x) (void) hipEventDestroy(x);
      c->err = std::string("upload pipeline: ") + hipGetErrorString(e);
      return BPVO_ERR_DEVICE;
    }
    c->up_pinned.push_back(pin);
    c->up_slot_free.push_back(ev[0]);
    c->up_slot_free.push_back(ev[1]);
  }
  if(n_pairs > c->up_cap_pairs) {
    HIP_CK(c, hipDeviceSynchronize());
    (void) hipFree(c->up_d_img); (void) hipFree(c->up_d_disp);
    c->up_d_img = nullptr; c->up_d_disp = nullptr; c->up_cap_pairs = 0;
    HIP_CK(c, hipMalloc((void**) &c->up_d_img, (size_t) 2 * n_pairs * npix));
    HIP_CK(c, hipMalloc((void**) &c->up_d_disp, (size_t) n_pairs * npix * sizeof(float)));
    c->up_cap_pairs = n_pairs;
  }
  return BPVO_OK;
}

// The pairs of a host batch are cut into nl * nsub groups of consecutive pairs, uploaded in that order; lane k runs the groups k, k + nl,
// k + 2 nl, ... one after the other, each end to end (frame stage as its chunks land, template, estimate).  With nsub = 1 a lane's first
// kernel of the Gauss-Newton stage waits for HALF the batch (2 lanes) to cross the bus and the second lane for all of it: 46 ms of a
// 196 ms step were exposed upload (profiles/r03_host_buffers_first.txt).  With nsub = 2 the first group is a quarter of the batch and
// every later group has landed long before its lane gets to it.
void host_groups(int n_pairs, int nl, int nsub, std::vector<std::pair<int, int>>& groups)
{
  const int ng = nl * nsub;
  groups.clear();
  for(int g = 0; g < ng; ++g) {
    const int lo = (int) ((long long) n_pairs * g / ng), hi = (int) ((long long) n_pairs * (g + 1) / ng);
    groups.emplace_back(lo, hi);
  }
}
// Two lanes, three groups in upload order: [0, a) -> lane 0, [a, a + b) -> lane 1, the rest -> lane 0 again (group 3, lane 1's second, is
// empty).  With two equal groups nothing but frame kernels runs for the first 30 ms of a 1024-pair step (lane 0's half has to land
// first: profiles/r03_host_timeline.txt); a first group of a fifth of the batch has landed after 10 ms, and what it loses as a small
// batch is less than the 17 ms it gains.
void host_groups_plan(const bpvo_hip_ctx* c, int n_pairs, std::vector<std::pair<int, int>>& groups)
{
  auto chunks = [](double pairs) { return (int) std::lround(pairs / kUploadChunkPairs) * kUploadChunkPairs; };
  const int a = std::max(4 * kUploadChunkPairs, chunks(c->up_plan[0] * n_pairs));
  const int b = std::max(4 * kUploadChunkPairs, std::min(n_pairs - a - 4 * kUploadChunkPairs, chunks(c->up_plan[1] * n_pairs)));
  groups.clear();
  groups.emplace_back(0, a);
  groups.emplace_back(a, a + b);
  groups.emplace_back(a + b, n_pairs);
  groups.emplace_back(n_pairs, n_pairs);
}
// starts the workers; chunks are cut inside the groups, in group order
int upload_start(bpvo_hip_ctx* c, UploadRun& u, int n_pairs, const std::vector<std::pair<int, int>>& groups, const uint8_t* images, const float* disparities)
{
  int rc = upload_prepare(c, n_pairs);
  if(rc) return rc;
  u.c = c; u.n_pairs = n_pairs;
  for(const auto& g : groups)
    for(int p0 = g.first; p0 < g.second; p0 += kUploadChunkPairs) u.chunks.emplace_back(p0, std::min(kUploadChunkPairs, g.second - p0));
  const int nchunks = (int) u.chunks.size();
  while((int) c->up_chunk_done.size() < nchunks) {
    hipEvent_t e = nullptr;
    HIP_CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->up_chunk_done.push_back(e);
  }
  u.recorded.assign(nchunks, 0);
  const size_t npix = c->geom[0].npix;
  const int T = std::min(c->up_workers, nchunks);
  const auto t_start = std::chrono::steady_clock::now();
  c->up_last_bytes = (size_t) n_pairs * npix * (2 + 4);
  auto done_count = std::make_shared<std::atomic<int>>(0);
  for(int w = 0; w < T; ++w) {
    u.workers.emplace_back([c, &u, w, T, nchunks, npix, images, disparities, t_start, done_count] {
      (void) hipSetDevice(c->device);
      hipStream_t st = c->up_streams[0];      // one copy stream for all workers (see bpvo_hip_ctx::up_workers)
      int turn = 0;
      for(int k = w; k < nchunks; k += T, ++turn) {
        const int p0 = u.chunks[k].first, np = u.chunks[k].second, sl = turn & 1;
        hipError_t e = hipEventSynchronize(c->up_slot_free[2 * w + sl]);       // the copy that last read this slot has finished
        uint8_t* pin_img = c->up_pinned[w] + (size_t) sl * c->up_slot_bytes;
        float* pin_disp = reinterpret_cast<float*>(pin_img + (size_t) kUploadChunkPairs * npix * 2);
        std::memcpy(pin_img, images + (size_t) 2 * p0 * npix, (size_t) 2 * np * npix);
        for(int i = 0; i < np; ++i) std::memcpy(pin_disp + (size_t) i * npix, disparities + (size_t) 2 * (p0 + i) * npix, npix * sizeof(float));
        if(e == hipSuccess) e = hipMemcpyAsync(c->up_d_img + (size_t) 2 * p0 * npix, pin_img, (size_t) 2 * np * npix, hipMemcpyHostToDevice, st);
        if(e == hipSuccess) e = hipMemcpyAsync(c->up_d_disp + (size_t) p0 * npix, pin_disp, (size_t) np * npix * sizeof(float), hipMemcpyHostToDevice, st);
        if(e == hipSuccess) e = hipEventRecord(c->up_slot_free[2 * w + sl], st);
        if(e == hipSuccess) e = hipEventRecord(c->up_chunk_done[k], st);
        {
          std::lock_guard<std::mutex> lk(u.mu);
          if(e != hipSuccess && u.err.empty()) u.err = std::string("upload pipeline: ") + hipGetErrorString(e);
          u.recorded[k] = 1;
        }
        u.cv.notify_all();
      }
      (void) hipStreamSynchronize(st);
      if(done_count->fetch_add(1) + 1 == T)
        c->up_last_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    });
  }
  return BPVO_OK;
}

// setData of the pairs [lo, hi) of a lane from the staging area as the chunks land, up_group chunks per frame-stage launch (the first
// group of a lane is a single chunk: the device starts as soon as anything is there)
int upload_consume(bpvo_hip_ctx* c, UploadRun& u, int lo, int hi, const FrameRun& fr_lane)
{
  const size_t npix = c->geom[0].npix;
  std::vector<size_t> mine;
  for(size_t k = 0; k < u.chunks.size(); ++k)
    if(u.chunks[k].first >= lo && u.chunks[k].first < hi) mine.push_back(k);
  for(size_t i = 0; i < mine.size();) {
    const size_t take = std::min(mine.size() - i, (size_t) (i == 0 ? 1 : kUploadGroup));
    for(size_t q = i; q < i + take; ++q) {
      const size_t k = mine[q];
      {
        std::unique_lock<std::mutex> lk(u.mu);
        u.cv.wait(lk, [&] { return u.recorded[k] != 0; });
        if(!u.err.empty()) { (fr_lane.own_thread ? fr_lane.ln->err : c->err) = u.err; return BPVO_ERR_DEVICE; }
      }
      FR_CK(c, fr_lane, hipStreamWaitEvent(fr_lane.stream, c->up_chunk_done[k], 0));
    }
    const int p0 = u.chunks[mine[i]].first;
    int np = 0;
    for(size_t q = i; q < i + take; ++q) np += u.chunks[mine[q]].second;      // (the chunks of a lane are contiguous)
    FrameRun fr = fr_lane;
    fr.tab = 2 * p0;
    fr.selected_ev = nullptr; fr.on_selected = nullptr;
    // skip_odd_disp = 2: the staging area holds the A frames' disparities only, packed
    int rc = frames_set_data(c, 2 * p0, 1, 2 * np, c->up_d_img + (size_t) 2 * p0 * npix, c->up_d_disp + (size_t) p0 * npix, true, fr, 2);
    if(rc) return rc;
    i += take;
  }
  return BPVO_OK;
}

// Staggered lanes (round 2).  A batch whose frame stage ran as a whole before any estimation starts every lane at the coarsest
// pyramid level at the same moment: for the first two levels (a few hundred points per pair) every launch is latency-bound and the
// chip idles, whatever the number of lanes.  Here each lane runs ITS pairs end to end on its own stream — setData, setTemplate,
// estimatePose — and lane k's frame stage is queued behind lane k-1's selection: the chip-filling frame kernels of one lane run
// under the narrow coarse-level iterations of the previous one, and the coarse levels of lane k under the fine levels of lane k-1.
// Same kernels on the same data per pair: results are bit-identical to the one-stage-at-a-time form (BPVO_HIP_STAGGER=0).
int batch_run_staggered(bpvo_hip_ctx* c, int n_pairs, int nl, const uint8_t* images, const float* disparities, bool on_device, float* poses,
                        bpvo_hip_stats* stats, UploadRun* pipe, const std::vector<std::pair<int, int>>* host_group_list)
{
  HIP_CK(c, hipStreamSynchronize(c->stream));
  c->frac_valid = false;
  const size_t npix = c->geom[0].npix;
  std::vector<int> rcs(nl, BPVO_OK);
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int> selected(nl, 0);     // 1: the lane recorded its selected_ev (or failed before: nobody waits for ever)
  // the groups of consecutive pairs a lane runs one after the other: one per lane, or (host buffers) nsub per lane in upload order
  std::vector<std::pair<int, int>> groups;
  if(host_group_list) groups = *host_group_list;
  else host_groups(n_pairs, nl, 1, groups);
  const int nsub = (int) groups.size() / nl;
  auto run = [&](int k) {
    Lane* ln = &c->lanes[k];
    (void) hipSetDevice(c->device);
    auto release_next = [&mu, &cv, &selected, k] { { std::lock_guard<std::mutex> lk(mu); selected[k] = 1; } cv.notify_all(); };
    struct Release { std::function<void()> f; ~Release() { f(); } } always{release_next};   // whatever happens, nobody waits for ever
    // device-resident inputs: lane k's frame stage starts behind lane k - 1's selection (stagger); host inputs arrive staggered anyway
    if(k > 0 && !pipe) {
      { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return selected[k - 1] != 0; }); }
      if(hipStreamWaitEvent(ln->stream, c->lanes[k - 1].selected_ev, 0) != hipSuccess) { ln->err = "hipStreamWaitEvent"; rcs[k] = BPVO_ERR_DEVICE; return; }
    }
    for(int sub = 0; sub < nsub; ++sub) {
      const int lo = groups[(size_t) sub * nl + k].first, hi = groups[(size_t) sub * nl + k].second, n = hi - lo;
      if(n <= 0) continue;
      FrameRun fr{ln->stream, ln, 2 * lo, true, nullptr, nullptr};
      int rc = pipe ? upload_consume(c, *pipe, lo, hi, fr)
                    : frames_set_data(c, 2 * lo, 1, 2 * n, images + (size_t) 2 * lo * npix, disparities + (size_t) 2 * lo * npix, on_device, fr, c->keep_current_disparity ? 0 : 1);
      if(rc) { rcs[k] = rc; return; }
      if(sub == 0) {
        fr.selected_ev = ln->selected_ev;     // recorded, and the next lane released, before the template stage waits for its point counts
        fr.on_selected = release_next;
      }
      rc = frames_set_template(c, 2 * lo, 2, n, fr);
      if(rc) { rcs[k] = rc; return; }
      std::vector<int> wss(n), refs(n), curs(n);
      for(int i = 0; i < n; ++i) { wss[i] = lo + i; refs[i] = 2 * (lo + i); curs[i] = 2 * (lo + i) + 1; }
      rc = estimate_group(c, ln, n, wss.data(), refs.data(), curs.data(), nullptr, poses ? poses + 16 * (size_t) lo : nullptr,
                          stats ? stats + (size_t) lo * c->L : nullptr, c->d_records + (size_t) kRecordFloats * lo, false);
      if(rc) { rcs[k] = rc; return; }
    }
  };
  {
    std::vector<std::thread> th;
    for(int k = 1; k < nl; ++k) th.emplace_back(run, k);
    run(0);
    for(auto& t : th) t.join();
  }
  for(int k = 0; k < nl; ++k)
    if(rcs[k]) { c->err = c->lanes[k].err; return rcs[k]; }
  resolve_events(c);
  for(int i = 0; i < n_pairs; ++i) {
    Workspace& w = c->ws[i];
    w.last_ref = 2 * i;
    w.last_cur = 2 * i + 1;
    w.last_level = c->params.maxTestLevel;
  }
  return BPVO_OK;
}

}  // namespace

extern "C" {

int bpvo_hip_batch_run(bpvo_hip_ctx* c, int n_pairs, const uint8_t* images, const float* disparities, int on_device,
                       float* poses, bpvo_hip_stats* stats)
{
  CHECK_CTX(c);
  if(n_pairs < 0 || 2 * n_pairs > c->n_frames || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "batch exceeds ctx capacity");
  (void) hipSetDevice(c->device);
  if(n_pairs > 0 && (!images || !disparities)) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image/disparity");
  // host buffers: batches of at least two chunks go through the upload pipeline
  // (keep_current_disparity: the staging area of the pipeline holds the A frames' disparities only; such batches take plain copies)
  const bool use_pipe = !on_device && c->up_workers > 0 && !c->keep_current_disparity && n_pairs >= 2 * kUploadChunkPairs;
  int nl = lanes_for(c, n_pairs, use_pipe ? 2 : 8);      // (the upload plan is a two-lane plan)
  if(nl < 0) return nl;
  if(team_serves(c, n_pairs)) nl = 1;
  if(c->stagger && n_pairs >= c->stagger_min_pairs && nl > 1 && !c->profile_all) {
    if(!use_pipe) return batch_run_staggered(c, n_pairs, nl, images, disparities, on_device != 0, poses, stats, nullptr, nullptr);
    // (groups of at least 64 pairs: smaller ones cost more in launch floors than their earlier start is worth)
    std::vector<std::pair<int, int>> groups;
    const int nsub = 1;
    const bool plan = nl == 2 && nsub == 1 && c->up_plan[0] > 0.0 && n_pairs >= 32 * kUploadChunkPairs;
    if(plan) host_groups_plan(c, n_pairs, groups);
    else host_groups(n_pairs, nl, nsub, groups);
    UploadRun pipe;
    int rcp = upload_start(c, pipe, n_pairs, groups, images, disparities);
    if(rcp) return rcp;
    c->ctl_by_kernel = true;
    rcp = batch_run_staggered(c, n_pairs, nl, images, disparities, false, poses, stats, &pipe, &groups);
    c->ctl_by_kernel = false;
    return rcp;
  }
  int rc;
  if(use_pipe) {
    UploadRun pipe;
    std::vector<std::pair<int, int>> groups;
    host_groups(n_pairs, 1, 1, groups);
    rc = upload_start(c, pipe, n_pairs, groups, images, disparities);
    if(rc) return rc;
    rc = upload_consume(c, pipe, 0, n_pairs, ctx_run(c));
  } else {
    rc = frames_set_data(c, 0, 1, 2 * n_pairs, images, disparities, on_device != 0, c->keep_current_disparity ? 0 : 1);
  }
  if(rc) return rc;
  // The estimation follows on the same stream (one lane) or behind a synchronisation of it (several): the template stage ends without a
  // host round trip of its own, and — unless the team kernel runs next with every level in one launch (more than team_split_max_pairs
  // pairs) — leaves the normalisation sums of the levels below the coarsest on the side stream, under the Gauss-Newton iterations of the
  // coarsest (a single pair per call: 0.15 of 3.2 ms)
  FrameRun fr = ctx_run(c);
  fr.no_final_sync = !c->profiling;
  // (the team kernel: in two launches then, estimate.hip.  Whether it serves the batch is only known once the templates are — dense ones take the
  // chain — so parameters that allow a dense level defer as the chain does: 8 dense pairs lost 11 - 43 % to the sums in front of their first
  // iteration; sparse team batches of 8 - 16 pairs lose 2 - 4 % to the deferral's three launches: scripts/path_sweep.py)
  fr.defer_finest_nrm = !team_serves(c, n_pairs) || n_pairs <= c->team_split_max_pairs || templates_may_be_dense(c);
  rc = frames_set_template(c, 0, 2, n_pairs, fr);
  if(rc == BPVO_OK) rc = bpvo_hip_batch_estimate(c, n_pairs, nullptr, poses, stats);
  // (an error on the way: nothing of this call stays in flight)
  if(c->nrm_pending) { (void) hipEventSynchronize(c->nrm_pending); c->nrm_pending = nullptr; }
  if(c->nrm_pending_finest) { (void) hipEventSynchronize(c->nrm_pending_finest); c->nrm_pending_finest = nullptr; }
  return rc;
}
int bpvo_hip_batch_result_records_device(bpvo_hip_ctx* c, const float** d_records, int* floats_per_pair)
{
  CHECK_CTX(c);
  *d_records = c->d_records;
  *floats_per_pair = kRecordFloats;
  return BPVO_OK;
}

int bpvo_hip_batch_copy_records_device(bpvo_hip_ctx* c, float* d_dst, int n_pairs)
{
  CHECK_CTX(c);
  if(!d_dst || n_pairs < 0 || n_pairs > c->n_pairs) return fail(c, BPVO_ERR_INVALID_ARG, "bad record copy");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(d_dst, c->d_records, sizeof(float) * kRecordFloats * (size_t) n_pairs, hipMemcpyDeviceToDevice, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}

}  // extern "C"
