// select_dev.h — device-side building blocks of the finish stage (no kernels here: this header is shared by the
// translation units that finish a search inside their own kernels, scan_fused.h, and by select.h's kernels):
// folds of sorted k' lists by bitonic wave merges, the block tree merge, the stored-row accessor of the fp32 rerank
// (GPU analogue of src/vectordb.py:282-314) and the final ranking by (distance, index).
#pragma once
#include "scan_gemv.h"

namespace vdbk {

// ---- kk <= 64: every partial list is one sorted entry per lane -> bitonic folds ----
// Fold lists [l0, l1) of one query (layout [list][kk]) into the wave's list.
__device__ __forceinline__ void fold_sorted_lists(WaveTopK<1>& tk, const float* __restrict__ in_s,
                                                  const int* __restrict__ in_i, int l0, int l1, int step, int kk_, int stride = 0) {
  const int lane = lane_id();
  const bool valid = lane < kk_;
  const int kk = stride ? stride : kk_;   // elements between two lists
  constexpr int kAhead = 8;  // issue the loads of up to 8 lists before the (dependent) merge chains
  for (int l = l0; l < l1; l += kAhead * step) {
    float cs[kAhead];
    int ci[kAhead];
#pragma unroll
    for (int t = 0; t < kAhead; ++t) {
      const int ll = l + t * step;
      const bool ok = valid && ll < l1;
      cs[t] = ok ? in_s[(size_t)ll * kk + lane] : -INFINITY;
      ci[t] = ok ? in_i[(size_t)ll * kk + lane] : kIdxSentinel;
    }
#pragma unroll
    for (int t = 0; t < kAhead; ++t)
      if (l + t * step < l1) tk.merge_sorted(cs[t], ci[t]);  // wave-uniform
  }
}

// Tree-merge the NW per-wave lists of a block through LDS; wave 0 ends with the result.
template <int NW>
__device__ __forceinline__ void block_tree_merge(WaveTopK<1>& tk, float* ls, int* li) {
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (w >= half && w < 2 * half) { ls[(w - half) * 64 + lane] = tk.s[0]; li[(w - half) * 64 + lane] = tk.id[0]; }
    __syncthreads();
    if (w < half) tk.merge_sorted(ls[w * 64 + lane], li[w * 64 + lane]);
    __syncthreads();
  }
}

__device__ __forceinline__ bool dist_better(float d, long long i, float od, long long oi) {
  return (d < od) || (d == od && i < oi);
}

// Final ranking of one query's m candidates (score, local idx) held in LDS:
// order by (distance asc, index asc) — the same key the cross-shard merge uses,
// so results do not depend on the number of shards.  Writes k outputs.
__device__ __forceinline__ void rank_and_write(const float* ls, const int* li, int m, int k, long long row_offset,
                                               long long* __restrict__ out_idx, float* __restrict__ out_dist) {
  for (int t = threadIdx.x; t < k; t += blockDim.x) { out_idx[t] = -1; out_dist[t] = INFINITY; }
  __syncthreads();
  for (int t = threadIdx.x; t < m; t += blockDim.x) {
    const int id = li[t];
    const float sc = ls[t];
    if (id == kIdxSentinel || !(sc == sc)) continue;
    const float d = 1.0f - sc;
    int rank = 0;
    for (int j = 0; j < m; ++j) {
      const int oj = li[j];
      const float sj = ls[j];
      if (oj == kIdxSentinel || !(sj == sj)) continue;
      rank += dist_better(1.0f - sj, oj, d, id) ? 1 : 0;
    }
    if (rank < k) { out_idx[rank] = (long long)id + row_offset; out_dist[rank] = d; }
  }
}

// four consecutive elements j*4 .. j*4+3 of stored row `id`: from the fp32 copy, or (a store without one) from the bf16 rows
__device__ __forceinline__ f32x4 stored_row4(const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_b16, int id, int ld, int j) {
  if (rows_f32) return reinterpret_cast<const f32x4*>(rows_f32 + (size_t)id * ld)[j];
  const uint2 u = reinterpret_cast<const uint2*>(rows_b16 + (size_t)id * ld)[j];
  f32x4 r;
  r.x = __uint_as_float(u.x << 16); r.y = __uint_as_float(u.x & 0xffff0000u);
  r.z = __uint_as_float(u.y << 16); r.w = __uint_as_float(u.y & 0xffff0000u);
  return r;
}

}  // namespace vdbk
