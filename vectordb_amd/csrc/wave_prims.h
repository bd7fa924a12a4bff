// wave_prims.h — 64-lane wavefront primitives for gfx950 (CDNA4).
//   - DPP wave reduction (row_shr / row_bcast; no LDS traffic)
//   - WaveTopK<E>: a sorted top-k list distributed over the lanes of one wave
//     (position p = e*64 + lane), the building block of the fused selection
//     that replaces chroma-hnswlib's searchKnn priority queue.
// Ordering everywhere: higher score first, ties by lower row index.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdbk {

constexpr int kWave = 64;
constexpr int kIdxSentinel = 0x7fffffff;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND_CTRL>
__device__ __forceinline__ float dpp_mov(float old_v, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old_v), __float_as_int(src), CTRL,
                                                    ROW_MASK, BANK_MASK, BOUND_CTRL));
}

// Sum over the 64 lanes; every lane returns the total (via readlane 63).
__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_mov<0x111, 0xf, 0xf, true>(0.f, x);   // row_shr:1
  x += dpp_mov<0x112, 0xf, 0xf, true>(0.f, x);   // row_shr:2
  x += dpp_mov<0x114, 0xf, 0xf, true>(0.f, x);   // row_shr:4
  x += dpp_mov<0x118, 0xf, 0xf, true>(0.f, x);   // row_shr:8  -> lane 15 of each row = row sum
  x += dpp_mov<0x142, 0xa, 0xf, false>(0.f, x);  // row_bcast:15 into rows 1,3
  x += dpp_mov<0x143, 0xc, 0xf, false>(0.f, x);  // row_bcast:31 into rows 2,3 -> lane 63 = total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

__device__ __forceinline__ bool cand_better(float s, int i, float ts, int ti) {
  return (s > ts) || (s == ts && i < ti);
}

// Sorted (best first) list of up to E*64 candidates held in registers of one wave.
// Only the first kk positions are meaningful; thr_* mirrors position kk-1.
template <int E>
struct WaveTopK {
  float s[E];
  int id[E];
  float thr_s;
  int thr_i;
  int kk;

  __device__ __forceinline__ void init(int kk_) {
    kk = kk_;
#pragma unroll
    for (int e = 0; e < E; ++e) { s[e] = -INFINITY; id[e] = kIdxSentinel; }
    thr_s = -INFINITY;
    thr_i = kIdxSentinel;
  }

  __device__ __forceinline__ bool accepts(float cs, int ci) const { return cand_better(cs, ci, thr_s, thr_i); }

  __device__ __forceinline__ void refresh_thr() {
    const int pe = (kk - 1) >> 6, pl = (kk - 1) & 63;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (e == pe) {
        thr_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[e]), pl));
        thr_i = __builtin_amdgcn_readlane(id[e], pl);
      }
  }

  // Insert a wave-uniform candidate (caller checked accepts()).
  __device__ __forceinline__ void insert(float ns, int ni) {
    const int lane = lane_id();
    float cs = ns;
    int ci = ni;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool beats = cand_better(cs, ci, s[e], id[e]);
      const unsigned long long m = __ballot(beats);
      if (m != 0ull) {  // wave-uniform
        const int pos = __builtin_ctzll(m);
        const float up_s = __shfl_up(s[e], 1, 64);
        const int up_i = __shfl_up(id[e], 1, 64);
        const float last_s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s[e]), 63));
        const int last_i = __builtin_amdgcn_readlane(id[e], 63);
        if (lane > pos) { s[e] = up_s; id[e] = up_i; }
        else if (lane == pos) { s[e] = cs; id[e] = ci; }
        cs = last_s;
        ci = last_i;
      }
    }
    refresh_thr();
  }

  // Offer one candidate per lane (valid lanes only).
  __device__ __forceinline__ void push_lanes(float cs, int ci, bool valid) {
    unsigned long long m = __ballot(valid && accepts(cs, ci));
    while (m != 0ull) {  // wave-uniform loop
      const int l = __builtin_ctzll(m);
      m &= m - 1ull;
      const float ns = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), l));
      const int ni = __builtin_amdgcn_readlane(ci, l);
      if (accepts(ns, ni)) insert(ns, ni);
    }
  }

  // Offer one candidate per lane when MANY are expected to qualify (a fresh list, a loose
  // threshold): more than 8 accepting lanes are sorted in registers (bitonic network, 21
  // compare-exchange stages, best first) and merged in one bitonic wave merge instead of
  // being inserted one by one.  Same result as push_lanes.  E == 1 only.
  __device__ __forceinline__ void push_lanes_bulk(float cs, int ci, bool valid) {
    static_assert(E == 1, "bitonic wave sort handles one entry per lane");
    const bool take = valid && accepts(cs, ci);
    const unsigned long long m = __ballot(take);
    if (__builtin_popcountll(m) <= 8) { push_lanes(cs, ci, valid); return; }   // wave-uniform
    const int lane = lane_id();
    float vs = take ? cs : -INFINITY;
    int vi = take ? ci : kIdxSentinel;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
      for (int j = k >> 1; j > 0; j >>= 1) {
        const float ps = __shfl_xor(vs, j, 64);
        const int pi = __shfl_xor(vi, j, 64);
        const bool pb = cand_better(ps, pi, vs, vi);
        const bool keep_better = ((lane & j) == 0) == ((lane & k) == 0);
        if (keep_better ? pb : !pb) { vs = ps; vi = pi; }
      }
    merge_sorted(vs, vi);
  }

  // Merge another *sorted* (best first) list of <= 64 entries, one per lane
  // (sentinels behind its end), into this one.  Bitonic: lane i keeps the better
  // of own[i] and other[63-i] — the 64 best of the union, as a bitonic sequence —
  // then 6 compare-exchange stages restore the order.  ~10x cheaper than 64
  // insertions.  E == 1 only (lists of up to 64).
  __device__ __forceinline__ void merge_sorted(float os, int oi) {
    static_assert(E == 1, "bitonic wave merge handles one entry per lane");
    const int lane = lane_id();
    const float rs = __shfl(os, 63 - lane, 64);
    const int ri = __shfl(oi, 63 - lane, 64);
    if (cand_better(rs, ri, s[0], id[0])) { s[0] = rs; id[0] = ri; }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
      const float ps = __shfl_xor(s[0], j, 64);
      const int pi = __shfl_xor(id[0], j, 64);
      const bool pb = cand_better(ps, pi, s[0], id[0]);
      const bool lower = (lane & j) == 0;      // lower lane of the pair keeps the better one
      if (lower ? pb : !pb) { s[0] = ps; id[0] = pi; }   // upper lane keeps the worse (lists never hold NaN)
    }
    refresh_thr();
  }

  // Write the first kk positions to (LDS or global) arrays.
  __device__ __forceinline__ void store(float* ds, int* di) const {
    const int lane = lane_id();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int p = e * 64 + lane;
      if (p < kk) { ds[p] = s[e]; di[p] = id[e]; }
    }
  }
};

}  // namespace vdbk
