// K7 exact median of |r| (robust scale), device side: bracketed and full radix selection of one workspace by one workgroup.
#pragma once
#include "gn_common.h"

namespace bpvo_hip {

// ------------------------------------------------------------------------------------------------------------------
// K7 median + robust scale.  reference: AutoScaleEstimator::estimateScale / ScaleEstimator (bpvo/mestimator.cc:452-490)
// and median() (bpvo/utils.h:224-252): sigma = (1.4826f * (1 + 5/(n-6))) * median(|r| : valid), n = C * #valid (size_t
// arithmetic), sigma < 1e-6 -> 1, recomputed only while |sigma - sigma_prev| > 1e-6 (Q5, Q6).
//
// The order statistics x[n/2] (and x[n/2-1] for even n) are EXACT; they are found by MSB radix selection on the bit
// pattern of |r| (monotone for non-negative floats), with two cursors (lo, hi) refined in lock-step.  Two paths:
//
//  bracketed (every linearisation of a level but the first): the median moves little between GN iterations, so the
//    bracket step fused into warp_residual (bracket_block) only COUNTS the keys below a bracket [lo, hi) around the
//    previous median and compacts the few keys inside it into per-block candidate segments; K7b (median_finish_kernel)
//    then selects among the candidates only.  If the wanted ranks fall outside the bracket the full path runs — the
//    result is exact either way; the bracket width adapts to the last observed change.
//  full (first linearisation of a level, bracket miss): 3 passes over all keys, bits [30:20], [19:9], [8:0], one
//    workgroup per workspace (1024 threads, or 512 in launches wider than the chip: median_finish_kernel) with LDS histograms
//    (privatised copies in pass 1 to cut same-bin atomic serialisation); keys surviving pass 1 are cached in LDS so pass 3
//    never touches HBM again.
constexpr int MED_THREADS = 1024;     // median_finish_kernel; the persistent kernel runs the same code with 512 (template parameter NT)
constexpr int MED_COPIES = 4;
constexpr int MED_BINS = 2048;
constexpr int MED_CACHE = 20480;

struct MedCursor { unsigned prefix; unsigned rank; };

template <int NT = 1024>
__device__ __forceinline__ unsigned block_excl_scan_1024(unsigned v, unsigned* s_wave /*[16]*/, unsigned& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned incl = wave_incl_scan_u32(v);      // (DPP: gn_common.h)
  __syncthreads();
  if(lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned woff = 0, tot = 0;
#pragma unroll
  for(int w = 0; w < NT / 64; ++w) {
    const unsigned t = s_wave[w];
    if(w < wave) woff += t;
    tot += t;
  }
  total = tot;
  return woff + incl - v;
}

// every thread owns BPT = 2048 / NT consecutive bins (2t, 2t+1 for 1024 threads) of a (<= 2048)-bin histogram, h[] their counts,
// excl the number of keys in the bins before them: find the bins holding ranks k_lo / k_hi
template <int BPT>
__device__ __forceinline__ void find_ranks(const unsigned (&h)[BPT], unsigned excl, unsigned k_lo, unsigned k_hi, MedCursor* out /*[2]*/)
{
  unsigned b = (unsigned) BPT * threadIdx.x, e = excl;
#pragma unroll
  for(int q = 0; q < BPT; ++q) {
    if(k_lo >= e && k_lo < e + h[q]) { out[0].prefix = b + q; out[0].rank = k_lo - e; }
    if(k_hi >= e && k_hi < e + h[q]) { out[1].prefix = b + q; out[1].rank = k_hi - e; }
    e += h[q];
  }
}

template <int C, int NT, typename F>
__device__ __forceinline__ void for_each_valid_key(const PairJob& j, F f)
{
  const int n = j.n;
  if constexpr(C == 8) {
    const float4* q = reinterpret_cast<const float4*>(j.r.get());
    constexpr int U = 4;   // points in flight per thread: all loads of a round are issued before any is consumed
    for(int base = threadIdx.x; base < n; base += NT * U) {
      unsigned char v[U];
      float4 a[U], b[U];
#pragma unroll
      for(int u = 0; u < U; ++u) {
        const int pt = base + u * NT;
        const bool in = pt < n;
        v[u] = in ? j.valid[pt] : (unsigned char) 0;
        a[u] = in ? load_stream(q + tile_index<2>(pt, 0)) : make_float4(0, 0, 0, 0);
        b[u] = in ? load_stream(q + tile_index<2>(pt, 1)) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for(int u = 0; u < U; ++u) {
        if(!v[u]) continue;
        const int pt = base + u * NT;
        f(__float_as_uint(a[u].x) & 0x7fffffffu, pt); f(__float_as_uint(a[u].y) & 0x7fffffffu, pt);
        f(__float_as_uint(a[u].z) & 0x7fffffffu, pt); f(__float_as_uint(a[u].w) & 0x7fffffffu, pt);
        f(__float_as_uint(b[u].x) & 0x7fffffffu, pt); f(__float_as_uint(b[u].y) & 0x7fffffffu, pt);
        f(__float_as_uint(b[u].z) & 0x7fffffffu, pt); f(__float_as_uint(b[u].w) & 0x7fffffffu, pt);
      }
    }
  } else if constexpr(C != 1) {   // generic C: point-major records; every channel of the record (C x the groups of a wide descriptor)
    const int CT = C * j.n_groups;
    const size_t PT = (size_t) j.pitch;
    for(int pt = threadIdx.x; pt < n; pt += NT) {
      if(!j.valid[pt]) continue;
      for(int c0 = 0; c0 < CT; c0 += C) {
#pragma unroll
        for(int c = 0; c < C; ++c) f(__float_as_uint(j.r[(size_t) pt * PT + c0 + c]) & 0x7fffffffu, pt);
      }
    }
  } else {
    constexpr int U = 4;      // 16 points in flight per thread: a dense level (NMS off: 300 k points) is 18 rounds of this, each one memory latency
    for(int base = threadIdx.x * 4; base < n; base += NT * 4 * U) {   // n is a multiple of 16
      uchar4 v[U];
      float4 a[U];
#pragma unroll
      for(int u = 0; u < U; ++u) {
        const int p4 = base + u * NT * 4;
        const bool in = p4 < n;
        v[u] = in ? *reinterpret_cast<const uchar4*>(j.valid + p4) : make_uchar4(0, 0, 0, 0);
        a[u] = in ? *reinterpret_cast<const float4*>(j.r + p4) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for(int u = 0; u < U; ++u) {
        const int p4 = base + u * NT * 4;
        if(v[u].x) f(__float_as_uint(a[u].x) & 0x7fffffffu, p4 + 0);
        if(v[u].y) f(__float_as_uint(a[u].y) & 0x7fffffffu, p4 + 1);
        if(v[u].z) f(__float_as_uint(a[u].z) & 0x7fffffffu, p4 + 2);
        if(v[u].w) f(__float_as_uint(a[u].w) & 0x7fffffffu, p4 + 3);
      }
    }
  }
}

// One refinement pass of the two-cursor radix select.  A key takes part in cursor X iff its bits above (shift + width)
// equal X.prefix; its digit is (key >> shift) & (2^width - 1), width <= 11.  On return the cursors carry the extended
// prefix and the rank inside the selected digit bin.  Block-wide (1024 threads); `src(f)` calls f(key) for every key.
template <int NT, typename Src>
__device__ __forceinline__ void refine_pass(Src&& src, unsigned shift, unsigned width, MedCursor& lo, MedCursor& hi, unsigned* hist_lo,
                                            unsigned* hist_hi, unsigned* s_wave, MedCursor* cur)
{
  const int tid = threadIdx.x;
  const bool split = lo.prefix != hi.prefix;
  const unsigned nbins = 1u << width, up = shift + width;
  __syncthreads();
  for(unsigned i = tid; i < nbins; i += NT) { hist_lo[i] = 0; hist_hi[i] = 0; }
  __syncthreads();
  src([&](unsigned key) {
    const unsigned top = (up >= 32u) ? 0u : (key >> up);
    const unsigned dg = (key >> shift) & (nbins - 1u);
    if(top == lo.prefix) atomicAdd(&hist_lo[dg], 1u);
    else if(split && top == hi.prefix) atomicAdd(&hist_hi[dg], 1u);
  });
  __syncthreads();
  unsigned dummy;
  constexpr int BPT = MED_BINS / NT;
  const bool own = (unsigned) BPT * tid < nbins;        // nbins is a power of two >= BPT or smaller than it: bins past nbins read as 0
  unsigned ha[BPT], hb[BPT], sa = 0, sb = 0;
#pragma unroll
  for(int q = 0; q < BPT; ++q) { ha[q] = (own && (unsigned) BPT * tid + q < nbins) ? hist_lo[BPT * tid + q] : 0u; sa += ha[q]; }
  const unsigned ea = block_excl_scan_1024<NT>(sa, s_wave, dummy);
  unsigned eb = ea;
  if(split) {
#pragma unroll
    for(int q = 0; q < BPT; ++q) { hb[q] = (own && (unsigned) BPT * tid + q < nbins) ? hist_hi[BPT * tid + q] : 0u; sb += hb[q]; }
    eb = block_excl_scan_1024<NT>(sb, s_wave, dummy);
  }
  MedCursor tmp[2];
  tmp[0].prefix = 0xffffffffu; tmp[1].prefix = 0xffffffffu; tmp[0].rank = tmp[1].rank = 0;
  if(own) {
    find_ranks<BPT>(ha, ea, lo.rank, split ? 0xffffffffu : hi.rank, tmp);
    if(tmp[0].prefix != 0xffffffffu) { cur[0].prefix = (lo.prefix << width) | tmp[0].prefix; cur[0].rank = tmp[0].rank; }
    if(!split && tmp[1].prefix != 0xffffffffu) { cur[1].prefix = (hi.prefix << width) | tmp[1].prefix; cur[1].rank = tmp[1].rank; }
    if(split) {
      tmp[1].prefix = 0xffffffffu;
      find_ranks<BPT>(hb, eb, 0xffffffffu, hi.rank, tmp);
      if(tmp[1].prefix != 0xffffffffu) { cur[1].prefix = (hi.prefix << width) | tmp[1].prefix; cur[1].rank = tmp[1].rank; }
    }
  }
  __syncthreads();
  lo = cur[0];
  hi = cur[1];
  __syncthreads();
}

// The work of one NT-thread workgroup on workspace j; `st` is the state it reads and (thread 0, at the very end) updates — the
// workspace's own in HBM, or a workgroup-local copy (persistent kernel, where every workgroup runs the selection redundantly and
// `stats` is true for one of them only).
// COPIES privatised pass-1 histograms (a power of two >= 2: the second one doubles as the segment-offset table of the bracketed path),
// CACHE words of key cache: the LDS footprint is ((COPIES + 1) * MED_BINS + CACHE + 24) words.
template <int C, int NT, int COPIES = MED_COPIES, int CACHE = MED_CACHE>
__device__ __forceinline__ void median_block(const PairJob& j, GNState* st, unsigned char* smem_raw, bool stats, bool dense = false)
{
  unsigned* hist_lo = reinterpret_cast<unsigned*>(smem_raw);              // [COPIES][MED_BINS]
  unsigned* hist_hi = hist_lo + COPIES * MED_BINS;                    // [MED_BINS]
  unsigned* cache = hist_hi + MED_BINS;                                   // [CACHE]
  unsigned* s_wave = cache + CACHE;                                   // [16]
  MedCursor* cur = reinterpret_cast<MedCursor*>(s_wave + 16);             // [2]
  unsigned* s_misc = reinterpret_cast<unsigned*>(cur + 2);                // [0] cache count, [1] first valid point

  const int tid = threadIdx.x;
  float median = 0.0f;
  unsigned n_total = 0;
  bool done = false;
  unsigned tap_hits = 0, tap_lookups = 0;      // tap-cache statistics of this linearisation's warp_residual pass (bracket counters)

  // ---- bracketed path
  if(st->median_valid) {
    // totals of the per-block counters written by the bracket step of warp_residual (bracket_block)
    // (wide descriptors: the chunks of every channel group, one run of segments after the other; the valid-point counters then add up to
    // groups x valid points, and C x that is every valid entry, as below)
    const int nblk = ((j.n + K6_BLOCK - 1) / K6_BLOCK) * j.n_groups;
    unsigned c_below = 0, c_in = 0, c_valid = 0, c_hit = 0, cnt_first = 0;    // cnt_first: candidates of segment `tid`
    if(!dense) {
      for(int b = tid; b < nblk; b += NT) {
        const uint4 o = reinterpret_cast<const uint4*>(j.med_blk.get())[b];
        if(b == tid) cnt_first = o.y;
        c_below += o.x; c_in += o.y; c_valid += o.z; c_hit += o.w;
      }
    }
    unsigned t_below, t_in, t_valid;
    unsigned run_incl = 0;      // dense form: inclusive scan of the runs' candidate counts (lane = run)
    if(dense) {   // the totals the chunks' leaders added up, run by run (bracket_chunk<C, true>); wave 0 clears them at the end
      static_assert(kDenseRuns == 64, "one run per lane");
      const uint4 o = reinterpret_cast<const uint4*>(j.med_blk.get())[j.med_tot + 8 * (tid & 63)];      // {inside, below, valid points, tap-cache hits}
      run_incl = wave_incl_scan_u32(o.x);
      t_in = (unsigned) __builtin_amdgcn_readlane((int) run_incl, 63);
      t_below = wave_sum_u32(o.y); t_valid = wave_sum_u32(o.z); tap_hits = wave_sum_u32(o.w);
      tap_lookups = j.tapcache_on ? t_valid : 0u;
    } else {   // four block sums with one LDS round
      c_below = wave_sum_u32(c_below); c_in = wave_sum_u32(c_in); c_valid = wave_sum_u32(c_valid); c_hit = wave_sum_u32(c_hit);
      __syncthreads();
      if((tid & 63) == 0) { unsigned* w4 = cache + (tid >> 6) * 4; w4[0] = c_below; w4[1] = c_in; w4[2] = c_valid; w4[3] = c_hit; }
      __syncthreads();
      t_below = t_in = t_valid = 0;
#pragma unroll
      for(int w = 0; w < NT / 64; ++w) { t_below += cache[w * 4 + 0]; t_in += cache[w * 4 + 1]; t_valid += cache[w * 4 + 2]; tap_hits += cache[w * 4 + 3]; }
      tap_lookups = j.tapcache_on ? t_valid : 0u;      // (no tap cache at dense levels: nothing looked up)
      __syncthreads();
    }
    const unsigned nt = (unsigned) C * t_valid, below = t_below, m = t_in;
    const unsigned lo_key = st->lo_key, range = st->hi_key - st->lo_key;
    const unsigned k_hi = nt / 2, k_lo = (nt % 2 == 0 && nt > 0) ? k_hi - 1 : k_hi;
    if(nt >= 3 && k_lo >= below && k_hi < below + m && range > 0) {
      MedCursor lo, hi;
      lo.prefix = 0; hi.prefix = 0; lo.rank = k_lo - below; hi.rank = k_hi - below;
      const unsigned nbits = 32u - (unsigned) __clz((int) range);        // offsets d = key - lo_key are < range < 2^nbits
      // The candidates sit in per-block segments of 256*C slots.  They are first gathered into LDS as one dense run (offsets d): a
      // flat index f over all candidates is mapped to (segment, slot) through the exclusive scan of the segment counts, so that every
      // thread has several independent loads in flight — walking the segments one after the other costs two dependent global
      // latencies per segment and wave, twice (histogram pass, ranking pass), which was most of this path's time.  Too many
      // candidates or segments for the LDS areas: the segment walk from global memory (same keys, same result).
      unsigned* s_off = hist_lo + MED_BINS;                              // [nblk + 1] — refine_pass only uses the first MED_BINS words of hist_lo
      constexpr unsigned kListRoom = 2u * (unsigned) NT;                  // cache[0 .. 2 NT): lists of the ranking step
      unsigned* dense_keys = cache + kListRoom;
      const bool in_lds = m <= (unsigned) CACHE - kListRoom && (dense || nblk < (COPIES - 1) * MED_BINS);
      const unsigned run_cap = dense_run_cap(j.n, C);
      if(dense) {      // s_off[run .. run + 1]: the run's place among all candidates
        if(tid < kDenseRuns) { s_off[tid + 1] = run_incl; if(tid == 0) s_off[0] = 0; }
        __syncthreads();
      }
      // the dense form's candidates: every wave walks its runs, each a contiguous array (four loads in flight per lane)
      auto for_each_dense = [&](auto f) {
        const int lane = tid & 63;
        for(int r = tid >> 6; r < kDenseRuns; r += NT / 64) {
          const unsigned o0 = s_off[r], cnt = s_off[r + 1] - o0;
          const unsigned* run = j.cand + (size_t) r * run_cap;
          const uint4* run4 = reinterpret_cast<const uint4*>(run);      // (a run starts on a multiple of 256 C words and is a multiple of four long)
          for(unsigned i0 = 4u * lane; i0 < cnt; i0 += 1024u) {
            uint4 v[4];
#pragma unroll
            for(int u = 0; u < 4; ++u) v[u] = (i0 + 256u * u < cnt) ? run4[(i0 + 256u * u) >> 2] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for(int u = 0; u < 4; ++u) {
              const unsigned i = i0 + 256u * u;
              if(i + 0 < cnt) f(v[u].x - lo_key, o0 + i + 0);
              if(i + 1 < cnt) f(v[u].y - lo_key, o0 + i + 1);
              if(i + 2 < cnt) f(v[u].z - lo_key, o0 + i + 2);
              if(i + 3 < cnt) f(v[u].w - lo_key, o0 + i + 3);
            }
          }
        }
      };
      if(in_lds && dense) {
        for_each_dense([&](unsigned d, unsigned at) { dense_keys[at] = d; });
        __syncthreads();
      } else if(in_lds) {
        unsigned run = 0;                                                  // running offset of the chunks of NT segments
        for(int b0 = 0; b0 < nblk; b0 += NT) {
          const int b = b0 + tid;
          const unsigned cnt = b0 == 0 ? cnt_first : (b < nblk ? reinterpret_cast<const uint4*>(j.med_blk.get())[b].y : 0u);
          unsigned total;
          const unsigned off = block_excl_scan_1024<NT>(cnt, s_wave, total);
          if(b < nblk) s_off[b] = run + off;
          run += total;
          __syncthreads();
        }
        if(tid == 0) s_off[nblk] = m;
        __syncthreads();
        constexpr int U = 4;
        int b = 0;                                                         // segment of the thread's current flat index (flat indices grow)
        for(unsigned f0 = tid; f0 < m; f0 += (unsigned) NT * U) {
          unsigned v[U];
#pragma unroll
          for(int u = 0; u < U; ++u) {
            const unsigned f = f0 + (unsigned) u * NT;
            if(f < m) {
              // first segment whose end lies beyond f: gallop, then bisect
              int step = 1, lo_b = b;
              while(lo_b + step < nblk && s_off[lo_b + step] <= f) { lo_b += step; step <<= 1; }
              int hi_b = min(lo_b + step, nblk);                           // s_off[lo_b] <= f < s_off[hi_b]
              while(hi_b - lo_b > 1) { const int mid = (lo_b + hi_b) >> 1; if(s_off[mid] <= f) lo_b = mid; else hi_b = mid; }
              b = lo_b;
              v[u] = j.cand[(size_t) b * K6_BLOCK * C + (f - s_off[b])];
            }
          }
#pragma unroll
          for(int u = 0; u < U; ++u) {
            const unsigned f = f0 + (unsigned) u * NT;
            if(f < m) dense_keys[f] = v[u] - lo_key;
          }
        }
        __syncthreads();
      }
      auto src = [&](auto f) {
        if(in_lds) {
          for(unsigned i = tid; i < m; i += NT) f(dense_keys[i]);
          return;
        }
        if(dense) {
          for_each_dense([&](unsigned d, unsigned) { f(d); });
          return;
        }
        const int lane = tid & 63, wave = tid >> 6;
        for(int b = wave; b < nblk; b += NT / 64) {
          const unsigned mb = reinterpret_cast<const uint4*>(j.med_blk.get())[b].y;
          const unsigned* seg = j.cand + (size_t) b * K6_BLOCK * C;
          for(unsigned i = lane; i < mb; i += 64) f(seg[i] - lo_key);
        }
      };
      unsigned remaining = nbits;
      bool first = true;
      while(remaining > 0) {
        const unsigned width = remaining > 11u ? 11u : remaining;
        const bool was_split = lo.prefix != hi.prefix;
        remaining -= width;
        refine_pass<NT>(src, remaining, width, lo, hi, hist_lo, hist_hi, s_wave, cur);
        if(!first || remaining == 0) continue;
        first = false;
        // After the first digit the selected bins usually hold a handful of keys: finish by direct ranking (each thread
        // ranks one key of the bin by counting the smaller ones) instead of more histogram passes.
        const unsigned dmask = (1u << width) - 1u;
        const unsigned n_lo = hist_lo[lo.prefix & dmask];
        const unsigned n_hi = was_split ? hist_hi[hi.prefix & dmask] : hist_lo[hi.prefix & dmask];
        if(n_lo > (unsigned) NT || n_hi > (unsigned) NT) continue;
        unsigned* list_lo = cache;                    // [NT]
        unsigned* list_hi = cache + NT;               // [NT]
        const bool same_bin = lo.prefix == hi.prefix;
        __syncthreads();
        if(tid == 0) { s_misc[0] = 0; s_misc[1] = 0; }
        __syncthreads();
        const unsigned p_lo = lo.prefix, p_hi = hi.prefix, sh = remaining;
        src([&](unsigned d) {
          const unsigned top = d >> sh;
          if(top == p_lo) list_lo[atomicAdd(&s_misc[0], 1u)] = d;
          else if(!same_bin && top == p_hi) list_hi[atomicAdd(&s_misc[1], 1u)] = d;
        });
        __syncthreads();
        // rank of list[t] = #{smaller} + #{equal with smaller index}; exactly one element has the wanted rank
        auto pick = [&](const unsigned* list, unsigned cnt, unsigned want, unsigned* out) {
          if((unsigned) tid < cnt) {
            const unsigned mine = list[tid];
            unsigned rk = 0;
            for(unsigned q = 0; q < cnt; ++q) {
              const unsigned o = list[q];
              rk += (o < mine || (o == mine && q < (unsigned) tid)) ? 1u : 0u;
            }
            if(rk == want) *out = mine;
          }
        };
        pick(list_lo, n_lo, lo.rank, &cur[0].prefix);
        if(same_bin) pick(list_lo, n_lo, hi.rank, &cur[1].prefix);
        else pick(list_hi, n_hi, hi.rank, &cur[1].prefix);
        __syncthreads();
        lo.prefix = cur[0].prefix; hi.prefix = cur[1].prefix;     // full offsets d now
        __syncthreads();
        remaining = 0;
      }
      const float v_lo = __uint_as_float(lo_key + lo.prefix), v_hi = __uint_as_float(lo_key + hi.prefix);
      median = (nt % 2 != 0) ? v_hi : (float) (((double) (v_lo + v_hi)) / 2.0);
      n_total = nt;
      done = true;
    }
  }

  // ---- full path
  if(!done) {
    for(int i = tid; i < (COPIES + 1) * MED_BINS; i += NT) hist_lo[i] = 0;
    if(tid == 0) { s_misc[0] = 0; s_misc[1] = 0xffffffffu; }
    __syncthreads();
    {   // pass 1: bits [30:20], privatised histogram copies
      unsigned* h = hist_lo + (tid & (COPIES - 1)) * MED_BINS;
      unsigned first = 0xffffffffu;
      for_each_valid_key<C, NT>(j, [&](unsigned key, int pt) {
        atomicAdd(&h[key >> 20], 1u);
        first = min(first, (unsigned) pt);
      });
      if(C == 1 && first != 0xffffffffu) atomicMin(&s_misc[1], first);
    }
    __syncthreads();
    constexpr int BPT = MED_BINS / NT;
    unsigned hh[BPT], hsum = 0;
#pragma unroll
    for(int q = 0; q < BPT; ++q) {
      hh[q] = 0;
#pragma unroll
      for(int c = 0; c < COPIES; ++c) hh[q] += hist_lo[c * MED_BINS + BPT * tid + q];
      hsum += hh[q];
    }
    const unsigned excl = block_excl_scan_1024<NT>(hsum, s_wave, n_total);
    if(n_total >= 3) {
      const unsigned k_hi = n_total / 2, k_lo = (n_total % 2 == 0) ? k_hi - 1 : k_hi;
      find_ranks<BPT>(hh, excl, k_lo, k_hi, cur);
      __syncthreads();
      MedCursor lo = cur[0], hi = cur[1];
      __syncthreads();
      // pass 2: bits [19:9] of the keys in the selected pass-1 bucket(s); survivors cached in LDS
      const unsigned p_lo = lo.prefix, p_hi = hi.prefix;
      refine_pass<NT>([&](auto f) {
        for_each_valid_key<C, NT>(j, [&](unsigned key, int) {
          const unsigned top = key >> 20;
          if(top == p_lo || top == p_hi) {
            const unsigned idx = atomicAdd(&s_misc[0], 1u);
            if(idx < CACHE) cache[idx] = key;
          }
          f(key);
        });
      }, 9u, 11u, lo, hi, hist_lo, hist_hi, s_wave, cur);
      const unsigned ncache = s_misc[0];
      // pass 3: bits [8:0]
      refine_pass<NT>([&](auto f) {
        if(ncache <= CACHE) { for(unsigned i = tid; i < ncache; i += NT) f(cache[i]); }
        else for_each_valid_key<C, NT>(j, [&](unsigned key, int) { f(key); });
      }, 0u, 9u, lo, hi, hist_lo, hist_hi, s_wave, cur);
      const float v_lo = __uint_as_float(lo.prefix), v_hi = __uint_as_float(hi.prefix);
      median = (n_total % 2 != 0) ? v_hi : (float) (((double) (v_lo + v_hi)) / 2.0);   // (*m + *middle) / 2.0, utils.h:236
    } else if(n_total > 0) {
      // median(): data.size() < 3 -> data[0] = first valid entry in channel-major order (Q5); only reachable for C == 1
      __syncthreads();
      const unsigned first = s_misc[1];
      median = (first != 0xffffffffu) ? fabsf(j.r[(size_t) first * C]) : 0.0f;
    }
  }

  if(dense && tid < kDenseRuns) reinterpret_cast<uint4*>(j.med_blk.get())[j.med_tot + 8 * tid] = make_uint4(0u, 0u, 0u, 0u);     // (every wave's reads lie before a barrier behind this one)
  if(tid == 0) {
    if(stats) {
      j.cnt[done ? 2 : 3] += 1ull;                                          // measurement: bracketed vs full selections
      // tap cache: a linearisation without bracket counters is the first of a level (keys reset: no hits, every valid point looks up)
      if(!st->median_valid) tap_lookups = j.tapcache_on ? n_total / (unsigned) C : 0u;
      j.cnt[5] += tap_hits; j.cnt[6] += tap_lookups;
      if(st->num_fun_evals < 8) { j.cnt[7] += tap_hits; j.cnt[8] += tap_lookups; }
    }
    const unsigned long long nm6 = (unsigned long long) n_total - 6ull;     // size_t wrap for n < 6 (Q5)
    float s = (1.4826f * (1.0f + 5.0f / (float) nm6)) * median;
    if((double) s < 1e-6) s = 1.0f;
    st->delta_scale = fabsf(s - st->scale);
    st->scale = s;
    // bracket for the next linearisation of this level: centred on this median, as wide as 2x the last relative
    // change + 1 % (first use: 25 %), at most 50 %
    if(n_total >= 3 && median > 0.0f) {
      float rel = 0.25f;
      // (gain 2.5 + 2 % until round 6: on the benched batch the bracket then held 5 - 15 % of the keys where the median moved by 0.02 - 1 % — 2 + 1 %
// holds half as many, misses 0.4 % of the selections instead of none, and is worth 0.5 % of the 1024-pair step and 2 % of a single pair's;
// 2 + 0.5 % misses more than it saves on a single pair: scripts/shard_ab.py with builds of -DMED_REL_FLOOR / -DMED_REL_GAIN)
#ifndef MED_REL_FLOOR
#define MED_REL_FLOOR 0.01f
#define MED_REL_GAIN 2.0f
#endif
      if(st->last_median > 0.0f) rel = fminf(0.5f, fmaxf(MED_REL_FLOOR, MED_REL_GAIN * fabsf(median - st->last_median) / st->last_median + MED_REL_FLOOR));
      st->last_median = median;
      st->lo_key = __float_as_uint(median * (1.0f - rel));
      st->hi_key = __float_as_uint(median * (1.0f + rel)) + 1u;
      st->median_valid = 1;
    } else {
      st->median_valid = 0;
    }
  }
}

// the second shape of median_finish_kernel (launches wider than the chip) and the LDS either shape needs
constexpr int MED_THREADS_B = 512, MED_COPIES_B = 2, MED_CACHE_B = 7168;   // 53 KB: three workgroups per CU
constexpr size_t kMedianLds = ((MED_COPIES + 1) * MED_BINS + MED_CACHE + 16 + 4 + 4) * sizeof(unsigned);
constexpr size_t kMedianLdsB = ((MED_COPIES_B + 1) * MED_BINS + MED_CACHE_B + 16 + 4 + 4) * sizeof(unsigned);

}  // namespace bpvo_hip
