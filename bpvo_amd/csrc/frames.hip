// libbpvo_hip, host side: the per-frame stages — VisualOdometryFrame::setData (pyramid, descriptors) and setTemplate (saliency, selection, points,
// normalisation, template pixels and gradients) for batches of frames — and the accessors of their results.
#include "host_ctx.h"

using namespace bpvo_hip;
using namespace bpvo_hip_host;

namespace bpvo_hip_host {

FrameRun ctx_run(bpvo_hip_ctx* c) { return FrameRun{c->stream, &c->lanes[0], 0, false, nullptr, nullptr}; }

// which: 0 = table of the setData stage, 1 = table of the setTemplate stage (two tables, so that queueing the template stage does not
// have to wait for the descriptor kernels that still read the first).  Returns the device table through *tab (row fr.tab of level 0).
int upload_frame_jobs(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr, int which, const FrameJob** tab)
{
  const size_t table = (size_t) which * c->L * c->n_frames;
  FR_CK(c, fr, hipEventSynchronize(fr.ln->staging_ev[which]));   // the pinned rows may still feed the copy of an earlier call
  for(int l = 0; l < c->L; ++l) {
    FrameJob* row = c->h_fjobs + table + (size_t) l * c->n_frames + fr.tab;
    for(int i = 0; i < count; ++i) row[i] = make_frame_job(c, c->frames[first + i * stride], l);
  }
  // rows [tab, tab + count) of every level in one copy
  const size_t pitch = sizeof(FrameJob) * (size_t) c->n_frames;
  static_assert(sizeof(FrameJob) % 8 == 0 && sizeof(PairJob) % 8 == 0, "copy_rows_kernel moves 8-byte words");
  // (small tables too — up to 512 frames, 0.4 MB: a 2-D copy of a few KB is a 20 us stop of the stream, the kernel 5 — what a stage of one or two frames notices)
  if(c->ctl_by_kernel.load() || count <= 512)
    launch_copy_rows(fr.stream, c->d_fjobs + table + fr.tab, c->h_fjobs + table + fr.tab, pitch, sizeof(FrameJob) * (size_t) count, c->L);
  else
    FR_CK(c, fr, hipMemcpy2DAsync(c->d_fjobs + table + fr.tab, pitch, c->h_fjobs + table + fr.tab, pitch, sizeof(FrameJob) * (size_t) count, (size_t) c->L,
                                  hipMemcpyHostToDevice, fr.stream));
  FR_CK(c, fr, hipEventRecord(fr.ln->staging_ev[which], fr.stream));
  *tab = c->d_fjobs + table + fr.tab;
  return BPVO_OK;
}

// VisualOdometryFrame::setData (reference: bpvo/vo_frame.cc:48-55) for `count` frames at once
// skip_odd_disp: the frames are the (A, B) frames of pairs, in that order: B (odd i) only ever serves as the CURRENT frame of its pair,
// whose disparity nothing reads (the reference copies what it is handed, bpvo/vo_frame.cc:50-51; estimatePose never looks at it) — it is
// neither uploaded nor copied: 40 % of a pair's input bytes
// (2: as 1, with the device-resident disparities packed for the even frames only — the staging area of the upload pipeline)
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device,
                    const FrameRun& fr, int skip_odd_disp)
{
  if(count <= 0) return BPVO_OK;
  const size_t npix = c->geom[0].npix;
  hipStream_t s = fr.stream;
  if(!on_device) {
    for(int i = 0; i < count; ++i) {
      FrameSlot& f = c->frames[first + i * stride];
      FR_CK(c, fr, hipMemcpyAsync(f.img[0], images + (size_t) i * npix, npix, hipMemcpyHostToDevice, s));
      if(!(skip_odd_disp && (i & 1)) && !fr.skip_disparity_upload)
        FR_CK(c, fr, hipMemcpyAsync(f.disp, disps + (size_t) i * npix, npix * sizeof(float), hipMemcpyHostToDevice, s));
    }
  }
  // pair batches: the compact channel-0 plane serves the saliency map of TEMPLATE frames only; the current frames' descriptor kernel
  // skips its store (the selection reads channel 0 from the records should such a frame be made a template later)
  for(int i = 0; i < count; ++i) c->frames[first + i * stride].ch0_valid = !(skip_odd_disp && (i & 1));
  // ... and the TEMPLATE frames (A, even) of a pair batch keep no records at the NMS levels (FrameSlot::lazy): bit-planes with the census
  // fused into the blur kernel, CD3 gradients.  Every other frame, and every frame set through the frame API, is dense.
  const bool lazy_ok = skip_odd_disp != 0 && c->lazy_template && c->params.descriptor == BPVO_DESC_BITPLANES && c->C == 8 &&
                       !(c->params.sigmaPriorToCensusTransform > 0.0f) && c->params.sigmaBitPlanes > 0.0f && c->params.gradientEstimation == BPVO_GRAD_CD3;
  for(int i = 0; i < count; ++i)
    for(int l = 0; l < c->L; ++l) c->frames[first + i * stride].lazy[l] = lazy_ok && !(i & 1) && c->geom[l].nms_radius > 0;
  const FrameJob* tab = nullptr;
  int rc = upload_frame_jobs(c, first, stride, count, fr, 0, &tab);
  if(rc) return rc;
  const int NF = c->n_frames;
  if(on_device) launch_ingest(s, tab, images, disps, npix, count, skip_odd_disp);   // one launch instead of 2 copies per frame
  {
    double px = 0;
    for(int l = 1; l < c->L; ++l) px += (double) c->geom[l].npix * count;
    ScopedTimer t(c, KC_PYRAMID, px, fr.ln);
    // ImagePyramid::compute (bpvo/image_pyramid.cc:43-50).  Few frames: up to three levels per launch (kernels_frame.hip pyramid_levels_kernel)
    bool grouped = count <= c->merge_levels_max_frames;
    for(int l = 0; l < c->L; ++l) grouped = grouped && c->geom[l].cols >= 8 && c->geom[l].rows >= 8;
    for(int l = 1; l < c->L;) {
      const int steps = grouped ? std::min(3, c->L - l) : 0;
      if(steps >= 2) {
        launch_pyramid_levels(s, tab + (size_t) (l - 1) * NF, NF, steps, c->geom[l + steps - 1].cols, c->geom[l + steps - 1].rows, count);
        l += steps;
      } else {
        launch_pyrdown(s, tab + (size_t) (l - 1) * NF, tab + (size_t) l * NF, c->geom[l].cols, c->geom[l].rows, count);
        l += 1;
      }
    }
  }
  {
    double px = 0;
    for(int l = c->L - 1; l >= c->params.maxTestLevel; --l) px += (double) c->geom[l].npix * count;
    ScopedTimer t(c, KC_DESCRIPTOR, px, fr.ln);
    // few frames, bit-planes with the census fused: every level in ONE launch (kernels_frame.hip level_job)
    const bool fused_bp = c->C == 8 && c->params.descriptor == BPVO_DESC_BITPLANES && !(c->params.sigmaPriorToCensusTransform > 0.0f) && c->params.sigmaBitPlanes > 0.0f;
    const int l_lo = c->params.maxTestLevel;
    // ... and the same for the census of the smoothed image + the bit-planes blur (conf/perf_bitplanes.cfg: two launches) and the intensity descriptor
    const bool smoothed_bp = c->C == 8 && c->params.descriptor == BPVO_DESC_BITPLANES && c->params.sigmaPriorToCensusTransform > 0.0f && c->params.sigmaBitPlanes > 0.0f;
    const bool plain_intensity = c->C == 1 && c->params.descriptor != BPVO_DESC_LAPLACIAN;
    const bool one_launch = (fused_bp || smoothed_bp || plain_intensity) && count <= c->merge_levels_max_frames && c->L - l_lo > 1;
    if(one_launch && fused_bp) {
      launch_bitplanes(s, tab + (size_t) l_lo * NF, c->geom[l_lo].cols, c->geom[l_lo].rows, count, c->params.sigmaBitPlanes, c->gauss_k, 1, c->L - l_lo, NF);
    } else if(one_launch && smoothed_bp) {
      launch_census(s, tab + (size_t) l_lo * NF, c->geom[l_lo].cols, c->geom[l_lo].rows, count, c->census_taps, c->L - l_lo, NF);
      launch_bitplanes(s, tab + (size_t) l_lo * NF, c->geom[l_lo].cols, c->geom[l_lo].rows, count, c->params.sigmaBitPlanes, c->gauss_k, 0, c->L - l_lo, NF);
    } else if(one_launch) {
      launch_intensity(s, tab + (size_t) l_lo * NF, c->geom[l_lo].cols, c->geom[l_lo].rows, count, c->L - l_lo, NF);
    }
    for(int l = c->L - 1; l >= c->params.maxTestLevel && !one_launch; --l) {   // DenseDescriptorPyramid::init (dense_descriptor_pyramid.cc:67-71)
      const FrameJob* jobs = tab + (size_t) l * NF;
      const LevelGeom& g = c->geom[l];
      if(c->params.descriptor == BPVO_DESC_CENTRAL_DIFFERENCE) {
        launch_central_difference(s, jobs, g.cols, g.rows, count, c->params.centralDifferenceRadius, c->cd_before, c->cd_after);
      } else if(c->params.descriptor == BPVO_DESC_LATCH) {
        launch_latch(s, jobs, g.cols, g.rows, count, c->params.latchNumBytes, c->params.latchHalfSsdSize, c->d_latch_off, c->latch_taps[0], c->latch_taps[1],
                     c->latch_after);
      } else if(c->C == 5 || c->C == 10) {
        launch_descriptor_fields(s, jobs, g.cols, g.rows, count, c->C == 10, c->df_g1, c->df_g2);
      } else if(c->C == 3) {
        launch_gradient_descriptor(s, jobs, g.cols, g.rows, count, c->grad_pre);
      } else if(c->C == 1) {
        if(c->params.descriptor == BPVO_DESC_LAPLACIAN) launch_laplacian(s, jobs, g.cols, g.rows, count, c->params.laplacianKernelSize);
        else launch_intensity(s, jobs, g.cols, g.rows, count);
      } else {
        // census fused into the bit-planes kernel unless the census is taken of the smoothed image or the planes stay unsmoothed
        const bool fused_census = !(c->params.sigmaPriorToCensusTransform > 0.0f) && c->params.sigmaBitPlanes > 0.0f;
        if(!fused_census)
          launch_census(s, jobs, g.cols, g.rows, count, c->params.sigmaPriorToCensusTransform > 0.0f ? c->census_taps : nullptr);
        launch_bitplanes(s, jobs, g.cols, g.rows, count, c->params.sigmaBitPlanes, c->gauss_k, fused_census ? 1 : 0);
      }
    }
  }
  FR_CK(c, fr, hipGetLastError());
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    f.has_data = true;
    f.has_disp = !(skip_odd_disp && (i & 1));
  }
  return BPVO_OK;
}
// the disparity of a slot whose data stage ran with FrameRun::skip_disparity_upload, from the caller's host buffer, on the context's copy stream; the
// context's stream waits for it (whatever is queued there later sees the disparity).  A copy from pageable memory holds the host until it is done
// (1.2 MB of a 640x480 disparity: 90 us): addFrame calls this with the estimate queued, so that the GPU works meanwhile.
int upload_disparity(bpvo_hip_ctx* c, int slot, const float* disparity)
{
  if(!c->copy_stream) {
    HIP_CK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_CK(c, hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
  }
  HIP_CK(c, hipMemcpyAsync(c->frames[slot].disp, disparity, c->geom[0].npix * sizeof(float), hipMemcpyHostToDevice, c->copy_stream));
  HIP_CK(c, hipEventRecord(c->copy_ev, c->copy_stream));
  HIP_CK(c, hipStreamWaitEvent(c->stream, c->copy_ev, 0));
  return BPVO_OK;
}
int frames_set_data(bpvo_hip_ctx* c, int first, int stride, int count, const uint8_t* images, const float* disps, bool on_device, int skip_odd_disp)
{
  if(count <= 0) return BPVO_OK;
  if(first < 0 || stride < 1 || first + (count - 1) * stride >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot range");
  if(!images || !disps) return fail(c, BPVO_ERR_INVALID_ARG, "nullptr image/disparity");
  return frames_set_data(c, first, stride, count, images, disps, on_device, ctx_run(c), skip_odd_disp);
}

// VisualOdometryFrame::setTemplate (reference: bpvo/vo_frame.cc:61-93 -> bpvo/template_data.cc:37-142) for `count` frames
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count, const FrameRun& fr)
{
  if(count <= 0) return BPVO_OK;
  hipStream_t s = fr.stream;
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    if(f.tmpl_slab) continue;
    size_t total = 0;
    FrameSlot tmp;
    carve_frame_tmpl(c, tmp, nullptr, &total);
    FR_CK(c, fr, hipMalloc(&f.tmpl_slab, total));
    FR_CK(c, fr, hipMemsetAsync(f.tmpl_slab, 0, total, s));
    carve_frame_tmpl(c, f, (unsigned char*) f.tmpl_slab, nullptr);
  }
  const FrameJob* tab = nullptr;
  int rc = upload_frame_jobs(c, first, stride, count, fr, 1, &tab);
  if(rc) return rc;
  const int NF = c->n_frames;
  int* const h_ints = c->h_ints + (size_t) fr.tab * kMaxLevels;
  int* const d_ints = c->d_ints + (size_t) fr.tab * kMaxLevels;
  const bpvo_hip_params& p = c->params;
  const int border = std::max(p.nonMaxSuppRadius, 3);   // template_data.cc:51
  // few frames, every level on the tiled path (NMS radius <= 1): the levels in ONE launch of each of its three kernels (kernels_frame.hip level_job)
  bool tiled = true;
  for(int l = p.maxTestLevel; l < c->L; ++l) tiled = tiled && c->geom[l].nms_radius <= 1;
  const bool one_launch = tiled && count <= c->merge_levels_max_frames && c->L - p.maxTestLevel > 1;
  hipEvent_t counts_ev = fr.ln ? fr.ln->round_ev[0] : nullptr;      // (the lane's round events are idle outside its estimation)
  // The normalisation — sequential sums in the reference's order, a latency chain of one workgroup per (frame, level): 0.2 ms whatever the
  // batch — is read by the Gauss-Newton kernels only (the Jacobian rows are rebuilt there; template_build stores pixels and gradients).
  // A stage that runs alone on the context's stream (single frames, batches on one lane) puts it on a stream of its own — forked right
  // behind the selection, next to the read-back of the point counts, the host's round trip for them and template_build — and joins
  // the two before it returns; lanes of a fanned-out batch keep it in line.
  hipStream_t side = nullptr;
  if(c->nrm_side_stream && !fr.own_thread && fr.ln == &c->lanes[0] && counts_ev) {
    if(!c->side_stream) {
      FR_CK(c, fr, hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
      FR_CK(c, fr, hipStreamCreateWithFlags(&c->side_stream2, hipStreamNonBlocking));
      for(auto& e : c->side_ev) FR_CK(c, fr, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    side = c->side_stream;
  }
  // ... and when the estimation follows on this stream within the same call (fr.defer_finest_nrm: a batch on one lane), only the COARSEST
  // level's sums are joined here: the others — the longest is the first level without non-maximum suppression, 100 k points of a
  // 1241x376 frame, 0.17 ms — are needed when the Gauss-Newton iterations leave the coarsest level: ctx->nrm_pending, waited for by the
  // estimation just before its second level (estimate.hip).
  const int with_nrm = c->dspace ? 0 : p.withNormalization;      // DisparitySpaceWarp::setNormalization is a no-op (bpvo/disparity_space_warp.h:87-90)
  const bool defer = side && fr.defer_finest_nrm && c->nrm_defer && c->L - p.maxTestLevel > 1 && !c->nrm_pending && !c->nrm_pending_finest;
  if(one_launch) {
    double px = 0;
    for(int l = p.maxTestLevel; l < c->L; ++l) px += (double) c->geom[l].npix * count;
    ScopedTimer t(c, KC_SALIENCY_SELECT, px, fr.ln);
    const LevelGeom& g = c->geom[p.maxTestLevel];
    launch_saliency_select(s, tab + (size_t) p.maxTestLevel * NF, c->C, g.cols, g.rows, count, 1, p.minSaliency, p.minValidDisparity, p.maxValidDisparity, border,
                           c->L - p.maxTestLevel, NF);
  }
  for(int l = c->L - 1; l >= p.maxTestLevel && !one_launch; --l) {
    const FrameJob* jobs = tab + (size_t) l * NF;
    const LevelGeom& g = c->geom[l];
    ScopedTimer t(c, KC_SALIENCY_SELECT, (double) g.npix * count, fr.ln);
    launch_saliency_select(s, jobs, c->C, g.cols, g.rows, count, g.nms_radius, p.minSaliency, p.minValidDisparity, p.maxValidDisparity, border);
  }
  if(side) {
    FR_CK(c, fr, hipEventRecord(c->side_ev[0], s));
    FR_CK(c, fr, hipStreamWaitEvent(side, c->side_ev[0], 0));
    // (timed like the in-line form, with events on the side stream: they are resolved once this stream — which the side stream joins — is synchronised)
    if(defer && templates_may_be_dense(c)) {
      // parameters that allow a dense level (NMS off on a large one): the finest level first, on its own stream (the longest chain of adds — 2 ms for
      // 300 k points — starts at once and runs on under the iterations of ALL the levels above it); the coarsest, then the levels between, on the other
      hipStream_t side2 = c->side_stream2;
      FR_CK(c, fr, hipStreamWaitEvent(side2, c->side_ev[0], 0));
      { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side2); launch_normalization(side2, tab, NF, count, p.maxTestLevel, p.maxTestLevel + 1, with_nrm, c->nrm_dpp_asm); }
      FR_CK(c, fr, hipEventRecord(c->side_ev[3], side2));
      c->nrm_pending_finest = c->side_ev[3];
      { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side); launch_normalization(side, tab, NF, count, c->L - 1, c->L, with_nrm, c->nrm_dpp_asm); }
      FR_CK(c, fr, hipEventRecord(c->side_ev[1], side));
      if(c->L - 1 > p.maxTestLevel + 1) {
        { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side); launch_normalization(side, tab, NF, count, p.maxTestLevel + 1, c->L - 1, with_nrm, c->nrm_dpp_asm); }
        FR_CK(c, fr, hipEventRecord(c->side_ev[2], side));
        c->nrm_pending = c->side_ev[2];
      }
    } else if(defer) {
      // (sparse templates: two launches on one stream — the third launch, the second stream and their events cost a single pair 1 % of its step)
      { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side); launch_normalization(side, tab, NF, count, c->L - 1, c->L, with_nrm, c->nrm_dpp_asm); }
      FR_CK(c, fr, hipEventRecord(c->side_ev[1], side));
      { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side); launch_normalization(side, tab, NF, count, p.maxTestLevel, c->L - 1, with_nrm, c->nrm_dpp_asm); }
      FR_CK(c, fr, hipEventRecord(c->side_ev[2], side));
      c->nrm_pending = c->side_ev[2];
    } else {
      { ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln, true, side); launch_normalization(side, tab, NF, count, p.maxTestLevel, c->L, with_nrm, c->nrm_dpp_asm); }
      FR_CK(c, fr, hipEventRecord(c->side_ev[1], side));
    }
  }
  // one read-back of the point counts: the host needs them to size the template-build and GN grids.  Queued AHEAD of an in-line normalisation
  // and waited for through an event of its own: the host's round trip (~30 us) then runs under the normalisation's sequential sums
  // instead of behind them.
  launch_gather_counts(s, tab, NF, count, p.maxTestLevel, c->L, d_ints);
  FR_CK(c, fr, hipMemcpyAsync(h_ints, d_ints, sizeof(int) * kMaxLevels * (size_t) count, hipMemcpyDeviceToHost, s));
  if(counts_ev) FR_CK(c, fr, hipEventRecord(counts_ev, s));
  if(fr.selected_ev) FR_CK(c, fr, hipEventRecord(fr.selected_ev, s));
  if(fr.on_selected) fr.on_selected();
  if(!side) {
    // the sequential (reference-order) normalisation sums of all levels and frames run side by side in one launch
    ScopedTimer t(c, KC_NORMALIZATION, 0.0, fr.ln);
    launch_normalization(s, tab, NF, count, p.maxTestLevel, c->L, with_nrm, c->nrm_dpp_asm);
  }
  if(counts_ev) FR_CK(c, fr, hipEventSynchronize(counts_ev));
  else FR_CK(c, fr, hipStreamSynchronize(s));
  std::vector<int> max_n(c->L, 0);
  double pts = 0;
  for(int i = 0; i < count; ++i) {
    FrameSlot& f = c->frames[first + i * stride];
    for(int l = 0; l < c->L; ++l) {
      f.n_host[l] = (l >= p.maxTestLevel) ? h_ints[(size_t) i * kMaxLevels + l] : 0;
      max_n[l] = std::max(max_n[l], f.n_host[l]);
      pts += f.n_host[l];
    }
  }
  if(c->profiling) {
    std::lock_guard<std::mutex> lk(c->units_mu);
    c->kc_units[KC_TEMPLATE] += pts;
    c->kc_units[KC_NORMALIZATION] += pts;
  }
  if(count <= c->merge_levels_max_frames && c->L - p.maxTestLevel > 1) {
    ScopedTimer t(c, KC_TEMPLATE, 0.0, fr.ln);
    int most = 0;
    for(int l = p.maxTestLevel; l < c->L; ++l) most = std::max(most, max_n[l]);
    launch_template_build(s, tab + (size_t) p.maxTestLevel * NF, c->C, most, count, p.gradientEstimation == BPVO_GRAD_CD5, c->gauss_k, c->L - p.maxTestLevel, NF);
  } else {
    for(int l = c->L - 1; l >= p.maxTestLevel; --l) {
      ScopedTimer t(c, KC_TEMPLATE, 0.0, fr.ln);
      launch_template_build(s, tab + (size_t) l * NF, c->C, max_n[l], count, p.gradientEstimation == BPVO_GRAD_CD5, c->gauss_k);
    }
  }
  if(side) FR_CK(c, fr, hipStreamWaitEvent(s, c->side_ev[1], 0));      // the normalisation joins here: whatever follows on this stream sees it
  if(!fr.own_thread && !fr.no_final_sync) {      // (a lane thread goes straight on to its estimation on the same stream)
    FR_CK(c, fr, hipStreamSynchronize(s));
    FR_CK(c, fr, hipGetLastError());
    resolve_events(c);
  }
  for(int i = 0; i < count; ++i) c->frames[first + i * stride].has_template = true;
  return BPVO_OK;
}
int frames_set_template(bpvo_hip_ctx* c, int first, int stride, int count)
{
  if(count <= 0) return BPVO_OK;
  if(first < 0 || stride < 1 || first + (count - 1) * stride >= c->n_frames) return fail(c, BPVO_ERR_INVALID_ARG, "bad frame slot range");
  for(int i = 0; i < count; ++i) {
    if(!c->frames[first + i * stride].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");   // vo_frame.cc:63
    if(!c->frames[first + i * stride].has_disp) return fail(c, BPVO_ERR_NO_DATA, "no disparity in frame (the current frame of a pair batch)");
  }
  return frames_set_template(c, first, stride, count, ctx_run(c));
}

// The full records of a slot's lazy levels, from the image the slot still holds: the descriptor kernel again, for this frame alone,
// with the flag cleared (accessors; a template frame of a batch used as the current frame of a later estimate).
int ensure_dense_descriptor(bpvo_hip_ctx* c, int slot)
{
  FrameSlot& f = c->frames[slot];
  bool any = false;
  for(int l = 0; l < c->L; ++l) any = any || f.lazy[l];
  if(!any) return BPVO_OK;
  // The job table is built from the flags, so they are cleared for it and put back if anything fails: the slot counts as dense
  // only once the kernels that fill its records have run.
  bool was[kMaxLevels];
  for(int l = 0; l < c->L; ++l) { was[l] = f.lazy[l]; f.lazy[l] = false; }
  auto restore = [&]() { for(int l = 0; l < c->L; ++l) f.lazy[l] = was[l]; };
  const FrameRun fr = ctx_run(c);
  const FrameJob* tab = nullptr;
  const int rc = upload_frame_jobs(c, slot, 1, 1, fr, 0, &tab);
  if(rc) { restore(); return rc; }
  for(int l = c->L - 1; l >= c->params.maxTestLevel; --l)
    launch_bitplanes(c->stream, tab + (size_t) l * c->n_frames, c->geom[l].cols, c->geom[l].rows, 1, c->params.sigmaBitPlanes, c->gauss_k, 1);
  if(hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    restore();
    return fail(c, BPVO_ERR_DEVICE, "ensure_dense_descriptor: descriptor kernels failed");
  }
  return BPVO_OK;
}

}  // namespace bpvo_hip_host

extern "C" {

int bpvo_hip_frame_set_data(bpvo_hip_ctx* c, int slot, const uint8_t* image, const float* disparity)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, slot, 1, 1, image, disparity, false);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));   // the caller may reuse its buffers on return (vo_frame.cc:50-51)
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frame_set_data_device(bpvo_hip_ctx* c, int slot, const uint8_t* d_image, const float* d_disparity)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, slot, 1, 1, d_image, d_disparity, true);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frames_set_data(bpvo_hip_ctx* c, int first_slot, int slot_stride, int count, const uint8_t* images,
                             const float* disparities, int on_device)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  int rc = frames_set_data(c, first_slot, slot_stride, count, images, disparities, on_device != 0);
  if(rc) return rc;
  HIP_CK(c, hipStreamSynchronize(c->stream));
  resolve_events(c);
  return BPVO_OK;
}
int bpvo_hip_frame_set_template(bpvo_hip_ctx* c, int slot)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  (void) hipSetDevice(c->device);
  return frames_set_template(c, slot, 1, 1);
}
int bpvo_hip_frames_set_template(bpvo_hip_ctx* c, int first_slot, int slot_stride, int count)
{
  CHECK_CTX(c);
  (void) hipSetDevice(c->device);
  return frames_set_template(c, first_slot, slot_stride, count);
}
int bpvo_hip_frame_clear(bpvo_hip_ctx* c, int slot)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  c->frames[slot].has_data = false;
  c->frames[slot].has_template = false;
  return BPVO_OK;
}
int bpvo_hip_frame_state(const bpvo_hip_ctx* c, int slot, int* has_data, int* has_template)
{
  if(!c || slot < 0 || slot >= c->n_frames) return BPVO_ERR_INVALID_ARG;
  *has_data = c->frames[slot].has_data;
  *has_template = c->frames[slot].has_template;
  return BPVO_OK;
}

// ---- accessors ------------------------------------------------------------------------------------------------------
int bpvo_hip_get_image(bpvo_hip_ctx* c, int slot, int level, uint8_t* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot);
  if(level < 0 || level >= c->L) return fail(c, BPVO_ERR_INVALID_ARG, "bad level");
  if(!c->frames[slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(out, c->frames[slot].img[level], c->geom[level].npix, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_descriptor_channel(bpvo_hip_ctx* c, int slot, int level, int channel, float* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  if(channel < 0 || channel >= c->C) return fail(c, BPVO_ERR_INVALID_ARG, "bad channel");
  if(!c->frames[slot].has_data) return fail(c, BPVO_ERR_NO_DATA, "no data in frame");
  (void) hipSetDevice(c->device);
  if(int rc = ensure_dense_descriptor(c, slot)) return rc;      // (a template frame of a pair batch: its NMS levels kept no records)
  const size_t npix = c->geom[level].npix;
  // de-interleave one channel: 2-D copy with a source pitch of C floats
  HIP_CK(c, hipMemcpy2DAsync(out, sizeof(float), c->frames[slot].desc[level] + channel, sizeof(float) * c->C, sizeof(float), npix,
                             hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_saliency(bpvo_hip_ctx* c, int slot, int level, float* out)
{
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);
  if(!c->frames[slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "no template");
  (void) hipSetDevice(c->device);
  HIP_CK(c, hipMemcpyAsync(out, c->frames[slot].sal[level], c->geom[level].npix * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
#define TMPL(c, slot, level)                                                                    \
  CHECK_CTX(c); CHECK_SLOT(c, slot); CHECK_LEVEL(c, level);                                     \
  if(!(c)->frames[slot].has_template) return fail(c, BPVO_ERR_NO_TEMPLATE, "no template");      \
  (void) hipSetDevice((c)->device);                                                             \
  FrameSlot& f = (c)->frames[slot];                                                             \
  const int n = f.n_host[level]

int bpvo_hip_num_points(bpvo_hip_ctx* c, int slot, int level, int* n_out) { TMPL(c, slot, level); *n_out = n; return BPVO_OK; }
int bpvo_hip_get_points(bpvo_hip_ctx* c, int slot, int level, float* xyzw)
{
  TMPL(c, slot, level);
  if(n) HIP_CK(c, hipMemcpyAsync(xyzw, f.pts[level], sizeof(float4) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_point_indices(bpvo_hip_ctx* c, int slot, int level, int* inds)
{
  TMPL(c, slot, level);
  if(n) HIP_CK(c, hipMemcpyAsync(inds, f.inds[level], sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  return BPVO_OK;
}
int bpvo_hip_get_pixels(bpvo_hip_ctx* c, int slot, int level, float* pixels)
{
  TMPL(c, slot, level);
  const int C = c->C;
  std::vector<float> t(tiled_floats(n, C));
  if(n) HIP_CK(c, hipMemcpyAsync(t.data(), f.pix[level], t.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  detile_to_channel_major(t.data(), n, C, 1, C == 8 ? 4 : C, pixels);
  return BPVO_OK;
}
int bpvo_hip_get_jacobians(bpvo_hip_ctx* c, int slot, int level, float* J)
{
  TMPL(c, slot, level);
  if(n == 0) return BPVO_OK;
  // the rows are not stored: evaluate them on the device from (point, Ix, Iy) exactly like irls_reduce does
  const size_t bytes = sizeof(float) * 6 * (size_t) n * c->C;
  float* d_out = nullptr;
  HIP_CK(c, hipMalloc((void**) &d_out, bytes));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  c->h_fjobs[0] = make_frame_job(c, f, level);
  hipError_t e = hipMemcpyAsync(c->d_fjobs, c->h_fjobs, sizeof(FrameJob), hipMemcpyHostToDevice, c->stream);
  if(e == hipSuccess) {
    launch_export_jacobians(c->stream, c->d_fjobs, c->C, n, d_out);
    e = hipMemcpyAsync(J, d_out, bytes, hipMemcpyDeviceToHost, c->stream);
  }
  if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void) hipFree(d_out);
  HIP_CK(c, e);
  return BPVO_OK;
}
int bpvo_hip_get_normalization(bpvo_hip_ctx* c, int slot, int level, float T[16], float T_inv[16])
{
  TMPL(c, slot, level);
  M44 t = m44_identity(), ti = m44_identity();
  // no normalisation was set (withNormalization off, an empty level, or DisparitySpaceWarp, whose setNormalization is a
  // no-op): the warp keeps the Identity it was constructed with (bpvo/rigid_body_warp.cc:27-28), not [1, -1 * 0]
  if(!c->params.withNormalization || c->dspace || n == 0) {
    std::memcpy(T, t.m, 64);
    std::memcpy(T_inv, ti.m, 64);
    return BPVO_OK;
  }
  float nrm[4];
  HIP_CK(c, hipMemcpyAsync(nrm, f.nrm + 4 * level, sizeof(nrm), hipMemcpyDeviceToHost, c->stream));
  HIP_CK(c, hipStreamSynchronize(c->stream));
  t.m[0] = t.m[5] = t.m[10] = nrm[0];
  t.m[3] = -nrm[0] * nrm[1]; t.m[7] = -nrm[0] * nrm[2]; t.m[11] = -nrm[0] * nrm[3];
  ti.m[0] = ti.m[5] = ti.m[10] = 1.0f / nrm[0];
  ti.m[3] = nrm[1]; ti.m[7] = nrm[2]; ti.m[11] = nrm[3];
  std::memcpy(T, t.m, 64);
  std::memcpy(T_inv, ti.m, 64);
  return BPVO_OK;
}

}  // extern "C"
