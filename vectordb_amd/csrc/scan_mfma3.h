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
// Main loop (one 512-thread block per CU, tile 256 corpus rows x 128 queries, K in 12
// steps of 64): 3-stage LDS ring filled by global_load_lds_dwordx4 with scalar base +
// per-thread constant 32-bit offsets (no per-step VALU address math), counted
// s_waitcnt vmcnt(6) + raw s_barrier so two stages stay in flight, all 16 fragment
// ds_read_b128 of a step issued before its 16 MFMAs, waves 4-7 issue their loads after
// their MFMAs (de-phases the two waves of a SIMD).
#pragma once
#include "scan_mfma.h"

namespace vdbk {

constexpr int kMf3DenseTiles = 16;                     // dense sample: 16 tiles = 4096 rows
constexpr int kMf3DenseLd = kMf3DenseTiles * kMfBM;

// MODE 0: dense sample scores -> dense[q][t*256 + r].  MODE 1: append hits to cand_*[q][cap].
template <int MODE>
__global__ __launch_bounds__(kMfThreads) void scan_mfma_pass_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap) {
  extern __shared__ __align__(16) unsigned char smem[];   // the only LDS object: 3 stages
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: keeps the wave-role branches and the staging cursor in SGPRs
  const int wm = w >> 1, wn = w & 1;
  const int hsel = lane >> 5;

  int qt, sp;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qt = j % n_qtiles;
    sp = xcd * (n_splits >> 3) + j / n_qtiles;
  } else {
    qt = blockIdx.x % n_qtiles;
    sp = blockIdx.x / n_qtiles;
  }
  const int qbase = qt * kMfBN;
  const int tile0 = sp * tiles_per_split;                               // pass-local tile index
  const int tile1 = min(tile0 + tiles_per_split, n_pass_tiles);
  const int my_tiles = tile1 > tile0 ? tile1 - tile0 : 0;
  const int total_steps = my_tiles * kMfSteps;
  if (total_steps == 0) return;

  // thresholds of this lane's two query columns (fixed for the whole pass)
  float thr[2] = {-INFINITY, -INFINITY};
  if (MODE == 1 && thr_in) {
    const int q0 = qbase + wn * 64 + (lane & 31);
    thr[0] = thr_in[min(q0, b - 1)];
    thr[1] = thr_in[min(q0 + 32, b - 1)];
  }
  // retire the threshold loads here: a first use inside the loop would make hipcc wait vmcnt(0)
  // (draining the two staged tiles in flight) at every tile's epilogue
  asm volatile("" :: "v"(thr[0]), "v"(thr[1]));

  // per-thread constant offsets: staging sources (bytes inside a tile) and fragment reads
  unsigned a_off[4], b_off[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int L = t * kMfThreads + tid, r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
    a_off[t] = (unsigned)(r * kMfRowBytes + c * 16);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int L = t * kMfThreads + tid, r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
    b_off[t] = (unsigned)(r * kMfRowBytes + c * 16);
  }
  const int ar0 = wm * 64 + (lane & 31), ar1 = ar0 + 32;
  const int br0 = wn * 64 + (lane & 31), br1 = br0 + 32;
  int fa0[4], fa1[4], fb0[4], fb1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 2 * s + hsel;
    fa0[s] = ar0 * 128 + ((c ^ ((ar0 >> 1) & 7)) << 4);
    fa1[s] = ar1 * 128 + ((c ^ ((ar1 >> 1) & 7)) << 4);
    fb0[s] = kMfStageA + br0 * 128 + ((c ^ ((br0 >> 1) & 7)) << 4);
    fb1[s] = kMfStageA + br1 * 128 + ((c ^ ((br1 >> 1) & 7)) << 4);
  }
  const int wbase16 = w * 1024;                       // this wave's 1 KiB slot inside a piece
  const unsigned char* qblk = qb16 + (size_t)qbase * kMfRowBytes;

  // staging cursor (scalar): next (tile, k-step) to issue
  int is_tile = 0, is_ks = 0;
  auto issue = [&](int buf) {
    const unsigned char* abase = rows + (size_t)(tile0 + is_tile) * tile_stride * (kMfBM * kMfRowBytes) + is_ks * (kMfBK * 2);
    const unsigned char* bbase = qblk + is_ks * (kMfBK * 2);
    unsigned char* st = smem + buf * kMfStage + wbase16;
#pragma unroll
    for (int t = 0; t < 4; ++t) mf_glds16(abase + a_off[t], st + t * (kMfThreads * 16));
#pragma unroll
    for (int t = 0; t < 2; ++t) mf_glds16(bbase + b_off[t], st + kMfStageA + t * (kMfThreads * 16));
    if (++is_ks == kMfSteps) { is_ks = 0; ++is_tile; }
  };

  f32x16_t acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  issue(0);
  if (total_steps > 1) issue(1);
  const bool late = w >= 4;

  int buf = 0, ks = 0, tile = 0;
  for (int step = 0; step < total_steps; ++step) {
    if (step + 1 < total_steps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool do_issue = step + 2 < total_steps;
    const int nbuf = buf == 0 ? 2 : buf - 1;                  // (step+2) % 3
    if (do_issue && !late) issue(nbuf);

    const unsigned char* sb = smem + buf * kMfStage;
    bf16x8_t a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a0[s] = *reinterpret_cast<const bf16x8_t*>(sb + fa0[s]);
      b0[s] = *reinterpret_cast<const bf16x8_t*>(sb + fb0[s]);
      a1[s] = *reinterpret_cast<const bf16x8_t*>(sb + fa1[s]);
      b1[s] = *reinterpret_cast<const bf16x8_t*>(sb + fb1[s]);
    }
    // pin the schedule: all 16 fragment reads are in flight before the first MFMA (hipcc otherwise
    // re-serialises them into 4 x [4 reads, wait, 4 MFMAs] on one register set, exposing LDS latency)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s], b0[s], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[s], b1[s], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], b0[s], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[s], b1[s], acc[1][1], 0, 0, 0);
    }
    if (do_issue && late) issue(nbuf);
    buf = buf == 2 ? 0 : buf + 1;
    if (++ks < kMfSteps) continue;
    ks = 0;

    // ---------------- epilogue of pass-local tile (tile0 + tile) ----------------
    const long long tile_row0 = (long long)(tile0 + tile) * tile_stride * kMfBM;
    const bool needs_mask = (tile_row0 + kMfBM > n_rows) || (qbase + kMfBN > b) || (self0 >= 0);
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {
      const int mt = idx >> 5, nt = (idx >> 4) & 1, reg = idx & 15;
      const float v = acc[mt][nt][reg];
      const int rloc = wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hsel;
      const int qg = qbase + wn * 64 + nt * 32 + (lane & 31);
      if (MODE == 0) {
        const int row = (int)tile_row0 + rloc;
        const bool ok = (row < n_rows) && !(self0 >= 0 && row == self0 + qg);
        if (qg < b) dense[(size_t)qg * kMf3DenseLd + (tile0 + tile) * kMfBM + rloc] = ok ? v : -INFINITY;
      } else {
        bool pass = v >= thr[nt];
        if (__ballot(pass) == 0ull) continue;          // the common case
        const int row = (int)tile_row0 + rloc;
        if (needs_mask) pass = pass && (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
        if (pass) {
          const int pos = atomicAdd(&cand_cnt[qg], 1);
          if (pos < cap) { cand_s[(size_t)qg * cap + pos] = v; cand_i[(size_t)qg * cap + pos] = row; }
        }
      }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    ++tile;
  }
}

// Per query: the k'-th best of its candidates -> thr_out[q]; optionally the sorted top-k'
// list -> list_*[q][kk].  Candidates: dense scores (cand_i == nullptr, index = position) or
// an append buffer with cand_cnt[q] entries (capacity cap; more -> *overflow = 1), optionally merged with a
// seed list (seed_*).
__global__ __launch_bounds__(1024) void select_kth_kernel(const float* __restrict__ cand_s, const int* __restrict__ cand_i,
                                                          const int* __restrict__ cand_cnt, int cap_or_n, int kk,
                                                          float* __restrict__ thr_out, float* __restrict__ list_s,
                                                          int* __restrict__ list_i, int* __restrict__ overflow,
                                                          const float* __restrict__ seed_s = nullptr,
                                                          const int* __restrict__ seed_i = nullptr) {
  __shared__ float ls[8 * 64];
  __shared__ int li[8 * 64];
  const int q = blockIdx.x;
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
  int n = cap_or_n;
  if (cand_cnt) {
    const int c = cand_cnt[q];
    if (c > cap_or_n) { if (threadIdx.x == 0) *overflow = 1; }
    else n = c;
  }
  const float* cs = cand_s + (size_t)q * cap_or_n;
  const int* ci = cand_i ? cand_i + (size_t)q * cap_or_n : nullptr;
  WaveTopK<1> tk;
  tk.init(kk);
  for (int base = w * 64; base < n; base += 1024) {
    const int i = base + lane;
    const bool valid = i < n;
    const float s = valid ? cs[i] : -INFINITY;
    const int id = valid ? (ci ? ci[i] : i) : kIdxSentinel;
    tk.push_lanes_bulk(s, id, valid && s > -INFINITY);
  }
  // per-wave lists are sorted: tree-merge with the bitonic wave merge
  {
    const int lanei = lane;
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
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
    if (list_s) tk.store(list_s + (size_t)q * kk, list_i + (size_t)q * kk);
  }
}

// Cheap valid threshold from the dense sample, one wave per query: lane l takes the maximum of the scores
// l, l+64, l+128, ...; the kk-th largest of the 64 lane maxima (kk <= 64) is the score of an actual sampled
// row with at least kk sampled rows at or above it, hence <= the true k'-th best.  A little looser than the
// exact k'-th best of the sample (1.5 % instead of 1 % of the next pass's scores pass at kk = 40 over 4096
// rows) but 10x cheaper than an exact selection over the dense scores (127 us at B = 1024).
__global__ __launch_bounds__(64) void dense_lower_bound_kernel(const float* __restrict__ scores, int ld, int n, int kk,
                                                               float* __restrict__ thr_out) {
  const int q = blockIdx.x;
  const int lane = threadIdx.x;
  const float* cs = scores + (size_t)q * ld;
  float v = -INFINITY;
  for (int i = lane; i < n; i += 64) v = fmaxf(v, cs[i]);
  // bitonic sort of the 64 maxima, descending by lane
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float o = __shfl_xor(v, j);
      v = (((lane & j) == 0) == ((lane & k) == 0)) ? fmaxf(v, o) : fminf(v, o);
    }
  const float t = __shfl(v, kk - 1);
  if (lane == 0) thr_out[q] = t;
}

}  // namespace vdbk
