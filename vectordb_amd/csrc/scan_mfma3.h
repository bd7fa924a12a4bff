// scan_mfma3.h — the production batched (B >= 16) bf16 kernel: Q.D^T on MFMA with a
// threshold/append epilogue.  (scan_mfma.h holds the exact in-kernel-list kernel that
// serves as the overflow fallback.)
//
// Why a second formulation: profiling the list kernel on MI355X (10M x 768, B=1024)
// showed 34 % of its time in the list epilogue (barriers, queue drains, threshold
// warm-up) and an issue-bound main loop.  Here the epilogue keeps NO state:
//   pass 0 (dense)  : scores of a small strided sample of corpus tiles are written out;
//                     a select kernel takes each query's k'-th best -> threshold t0.
//   pass i (append) : rows with score >= t(i-1) are appended to a per-query candidate
//                     buffer in HBM (one returning atomic per hit, hits are ~1e-4 of
//                     scores); select -> k'-th best -> t(i), over a ~100x larger sample.
//   last pass       : the whole corpus against t; select -> the exact top-k' lists.
// Any row of the true top-k' has score >= every t(i) (each t(i) is the k'-th best of a
// SUBSET), so the result is exact; if a candidate buffer would overflow (adversarial
// ordering), a device-side flag routes the batch through the exact list kernel instead.
// Extra work: the samples add ~1-2 % to the contraction.
//
// This header holds the protocol's selection kernels; the pass kernels themselves are
// scan_mfma5.h / scan_mfma6.h (256 x 256 and 256 x 128 tiles) and
// scan_stream.h (<= 64 queries, corpus straight from HBM into MFMA fragments).  The first
// pass kernel of this protocol (a barrier ring on a 256 x 128 tile) was retired in round 2:
// no shape was left where it won.
#pragma once
#include "scan_mfma.h"

namespace vdbk {

constexpr int kMf3DenseTiles = 128;                    // dense sample: at most 128 tiles = 32768 rows (the search picks 64 or 128 by store size; round 2: 16)
constexpr int kMf3DenseLd = kMf3DenseTiles * kMfBM;

#ifdef VDB_MFMA_HOST_TU   // kernels of the orchestrating TU (k_mfma.hip) only: every pass-kernel TU includes this header
// Per query: the k'-th best of its candidates -> thr_out[q]; optionally the sorted top-k'
// list -> list_*[q][kk].  Candidates: dense scores (cand_i == nullptr, index = position) or
// an append buffer with cand_cnt[q] entries (capacity cap; more -> *overflow = 1), optionally merged with a
// seed list (seed_*).
// zero_cnt != 0: the query's append counter is cleared for the next pass once every thread has read it.
// NW waves per query: the kernel is a chain of wave sorts and merges (latency, not bandwidth), so what counts is how many
// queries a CU holds at once and how deep the merge tree is - see the launch site for the measured choice.
template <int NW>
__global__ __launch_bounds__(NW * 64) void select_kth_kernel(const float* __restrict__ cand_s, const int* __restrict__ cand_i,
                                                          int* cand_cnt, int zero_cnt, int cap_or_n, int kk,
                                                          float* __restrict__ thr_out, float* __restrict__ list_s,
                                                          int* __restrict__ list_i, int* __restrict__ overflow,
                                                          const float* __restrict__ seed_s = nullptr,
                                                          const int* __restrict__ seed_i = nullptr) {
  __shared__ float ls[(NW / 2) * 64];
  __shared__ int li[(NW / 2) * 64];
  const int q = blockIdx.x;
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
  int n = cap_or_n;
  if (cand_cnt) {
    const int c = cand_cnt[q * kMfCntStride];
    if (c > cap_or_n) { if (threadIdx.x == 0) *overflow = 1; }
    else n = c;
  }
  const float* cs = cand_s + (size_t)q * cap_or_n;
  const int* ci = cand_i ? cand_i + (size_t)q * cap_or_n : nullptr;
  WaveTopK<1> tk;
  tk.init(kk);
  // kSelAhead chunks of 64 candidates per wave are loaded together (unconditional loads at clamped positions: a branch
  // around a load makes the compiler wait for each on its own, and the loop was a chain of ~1 us round trips — 25 us
  // for the 5 000 candidates per query of a sample pass at an 8-GPU shard's size)
  constexpr int kSelAhead = 8;
  for (int base0 = w * 64; base0 < n; base0 += kSelAhead * NW * 64) {
    float ps[kSelAhead];
    int pi[kSelAhead];
#pragma unroll
    for (int u = 0; u < kSelAhead; ++u) {
      const int i = min(base0 + u * NW * 64 + lane, n - 1);
      ps[u] = cs[i];
      pi[u] = ci ? ci[i] : i;
    }
#pragma unroll
    for (int u = 0; u < kSelAhead; ++u) {
      const int base = base0 + u * NW * 64;
      if (base >= n) break;   // wave-uniform
      const bool valid = base + lane < n;
      tk.push_lanes_bulk(valid ? ps[u] : -INFINITY, valid ? pi[u] : kIdxSentinel, valid && ps[u] > -INFINITY);
    }
  }
  // per-wave lists are sorted: tree-merge with the bitonic wave merge
  {
    const int lanei = lane;
#pragma unroll
    for (int half = NW / 2; half >= 1; half >>= 1) {
      if (w >= half && w < 2 * half) { ls[(w - half) * 64 + lanei] = tk.s[0]; li[(w - half) * 64 + lanei] = tk.id[0]; }
      __syncthreads();
      if (w < half) tk.merge_sorted(ls[w * 64 + lanei], li[w * 64 + lanei]);
      __syncthreads();
    }
  }
  if (w == 0) {
    // seed: a sorted list of kk entries per query from an earlier pass over OTHER rows (the sample's tiles
    // when the last pass skips them); -inf / sentinel entries behind its end
    if (seed_s) tk.merge_sorted(lane < kk ? seed_s[(size_t)q * kk + lane] : -INFINITY, lane < kk ? seed_i[(size_t)q * kk + lane] : kIdxSentinel);
    if (lane == 0) thr_out[q] = tk.thr_s;
    if (lane == 0 && zero_cnt && cand_cnt) cand_cnt[q * kMfCntStride] = 0;   // (behind the tree's barriers: every thread has its count)
    if (list_s) tk.store(list_s + (size_t)q * kk, list_i + (size_t)q * kk);
  }
}

// Cheap valid threshold from the dense sample, one wave per query: the kk-th largest of strided group maxima is the
// score of an actual sampled row with at least kk sampled rows at or above it, hence <= the true k'-th best.  A little
// looser than the exact k'-th best of the sample but ~10x cheaper than an exact selection over the dense scores
// (127 us at B = 1024).  Also clears the query's append counter for the first append pass.
__global__ __launch_bounds__(256) void dense_lower_bound_kernel(const float* __restrict__ scores, int ld, int n, int kk,
                                                                float* __restrict__ thr_out, int* __restrict__ cand_cnt) {
  // kk-th largest of 256 strided group maxima (4 per lane; groups of 16 at n = 4096, of 64 at n = 16384): about 1.06 % of
  // a 4096-row sample's scores lie above it, against 0.98 % for the exact kk-th best of the sample (kk = 40) - and 1.5 % for
  // the 64 group maxima this kernel used first: a third fewer hits in the sample pass, which is bound by its appends.
  // Four waves per query: wave w reads the 256-column slices w, w + 4, ... (n is a whole number of them: tiles of 256 rows),
  // all its loads in flight together (unconditional, at clamped slice numbers: a repeated slice does not change a maximum;
  // the one-wave form with a bounds test per load was a chain of round trips, 18 us for 32 queries), and wave 0 folds
  // the four partial maxima and bisects.
  constexpr int G = 4, kSlicesPerWave = kMf3DenseTiles / 4;
  __shared__ float part[3][G][64];
  const int q = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* cs = scores + (size_t)q * ld;
  const int n_slices = n >> 8;
  float v[G];
#pragma unroll
  for (int g = 0; g < G; ++g) v[g] = -INFINITY;
  if (n_slices > 0) {
    float x[kSlicesPerWave][G];
#pragma unroll
    for (int t = 0; t < kSlicesPerWave; ++t) {
      const int sl = min(w + 4 * t, n_slices - 1);
#pragma unroll
      for (int g = 0; g < G; ++g) x[t][g] = cs[sl * 256 + 64 * g + lane];
    }
#pragma unroll
    for (int t = 0; t < kSlicesPerWave; ++t)
#pragma unroll
      for (int g = 0; g < G; ++g) v[g] = fmaxf(v[g], x[t][g]);
  }
  for (int i = (n_slices << 8) + threadIdx.x; i < n; i += 256) v[0] = fmaxf(v[0], cs[i]);   // (a ragged tail: not produced by the protocol, kept for safety)
  if (w > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) part[w - 1][g][lane] = v[g];
  }
  __syncthreads();
  if (w > 0) return;
#pragma unroll
  for (int g = 0; g < G; ++g) v[g] = fmaxf(fmaxf(v[g], part[0][g][lane]), fmaxf(part[1][g][lane], part[2][g][lane]));
  // keys in unsigned order; the kk-th largest by bisection over the 32 key bits (counts by ballot)
  unsigned u[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const unsigned bits = __float_as_uint(v[g]);
    u[g] = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
  }
  unsigned res = 0;
#pragma unroll 1
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned cand = res | (1u << bit);
    int cnt = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) cnt += __popcll(__ballot(u[g] >= cand));
    if (cnt >= kk) res = cand;
  }
  // fewer than kk keys at all (a store smaller than kk rows): res stays 0, below every key
  const unsigned rb = (res & 0x80000000u) ? (res & 0x7fffffffu) : ~res;
  const float t = res == 0 ? -INFINITY : __uint_as_float(rb);
  if (lane == 0) { thr_out[q] = t; cand_cnt[q * kMfCntStride] = 0; }
}
#endif

}  // namespace vdbk
