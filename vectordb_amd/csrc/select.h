// select.h — small kernels after the scan: hierarchical merge of partial
// top-k lists, fp32 rerank (GPU analogue of src/vectordb.py:282-314), final
// ranking / output conversion, and the cross-shard merge used after the RCCL
// all-gather.
#pragma once
#include "scan_gemv.h"
#include "select_dev.h"

namespace vdbk {

constexpr int kMergeThreads = 256;
constexpr int kMergeWaves = kMergeThreads / 64;

// Merge candidates [q][n_cand] (any order; sentinel entries allowed) into
// gridDim.x lists of kk per query: out [q][gridDim.x][kk].
template <int E>
__global__ __launch_bounds__(kMergeThreads) void merge_lists_kernel(const float* __restrict__ in_s,
                                                                    const int* __restrict__ in_i, int n_cand,
                                                                    int cand_per_block, int kk,
                                                                    float* __restrict__ out_s,
                                                                    int* __restrict__ out_i) {
  const int q = blockIdx.y;
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
  const int start = blockIdx.x * cand_per_block;
  const int end = min(n_cand, start + cand_per_block);
  const float* cs_base = in_s + (size_t)q * n_cand;
  const int* ci_base = in_i + (size_t)q * n_cand;
  WaveTopK<E> tk[1];
  tk[0].init(kk);
  for (int base = start + w * 64; base < end; base += kMergeThreads) {
    const int i = base + lane;
    const bool valid = i < end;
    const float cs = valid ? cs_base[i] : -INFINITY;
    const int ci = valid ? ci_base[i] : kIdxSentinel;
    tk[0].push_lanes(cs, ci, valid);
  }
  // reuse the scan's block merge: writes out[q_local=0][part][kk] relative to the query base
  block_merge_store<1, E>(tk, kk, out_s + (size_t)q * gridDim.x * kk, out_i + (size_t)q * gridDim.x * kk,
                          gridDim.x, blockIdx.x);
}

// Level-1 merge: block (256 threads) folds `lists_per_block` lists; out [q][gridDim.x][kk].
__global__ __launch_bounds__(256) void merge_sorted_lists_kernel(const float* __restrict__ in_s,
                                                                 const int* __restrict__ in_i, int n_lists,
                                                                 int lists_per_block, int kk,
                                                                 float* __restrict__ out_s, int* __restrict__ out_i) {
  __shared__ float ls[2 * 64];
  __shared__ int li[2 * 64];
  const int q = blockIdx.y;
  const int w = threadIdx.x >> 6;
  const int l0 = blockIdx.x * lists_per_block;
  const int l1 = min(n_lists, l0 + lists_per_block);
  WaveTopK<1> tk;
  tk.init(kk);
  fold_sorted_lists(tk, in_s + (size_t)q * n_lists * kk, in_i + (size_t)q * n_lists * kk, l0 + w, l1, 4, kk);
  block_tree_merge<4>(tk, ls, li);
  if (w == 0) tk.store(out_s + ((size_t)q * gridDim.x + blockIdx.x) * kk, out_i + ((size_t)q * gridDim.x + blockIdx.x) * kk);
}

// No-rerank finish: merged [b][kk] -> outputs [b][k].  LDS: kk*(4+4) bytes.
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ m_s, const int* __restrict__ m_i,
                                                       int kk, int k, long long row_offset,
                                                       long long* __restrict__ out_idx,
                                                       float* __restrict__ out_dist) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* ls = reinterpret_cast<float*>(smem_raw);
  int* li = reinterpret_cast<int*>(ls + kk);
  const int q = blockIdx.x;
  for (int t = threadIdx.x; t < kk; t += blockDim.x) { ls[t] = m_s[(size_t)q * kk + t]; li[t] = m_i[(size_t)q * kk + t]; }
  __syncthreads();
  rank_and_write(ls, li, kk, k, row_offset, out_idx + (size_t)q * k, out_dist + (size_t)q * k);
}

// fp32 rerank: exact fp32 dot of the query with each of the kk candidate rows
// (fp32 copy of the store; rows_f32 == nullptr: the stored bf16 rows), then final ranking.  One block per query.
__global__ __launch_bounds__(256) void rerank_kernel(const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_b16, int ld,
                                                     const float* __restrict__ qn, const int* __restrict__ m_i,
                                                     int kk, int k, long long row_offset,
                                                     long long* __restrict__ out_idx,
                                                     float* __restrict__ out_dist) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* ls = reinterpret_cast<float*>(smem_raw);
  int* li = reinterpret_cast<int*>(ls + kk);
  const int q = blockIdx.x;
  const int w = threadIdx.x >> 6, nwv = blockDim.x >> 6;
  const int lane = lane_id();
  const f32x4* qv = reinterpret_cast<const f32x4*>(qn + (size_t)q * ld);
  const int c4 = ld >> 2;
  for (int c = w; c < kk; c += nwv) {
    const int id = m_i[(size_t)q * kk + c];
    float tot = -INFINITY;
    if (id != kIdxSentinel) {  // wave-uniform
      float acc = 0.f;
      for (int j = lane; j < c4; j += 64) {
        const f32x4 a = stored_row4(rows_f32, rows_b16, id, ld, j), b = qv[j];
        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
        acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
      }
      tot = wave_sum(acc);
    }
    if (lane == 0) { ls[c] = tot; li[c] = id; }
  }
  __syncthreads();
  rank_and_write(ls, li, kk, k, row_offset, out_idx + (size_t)q * k, out_dist + (size_t)q * k);
}

// kk <= 64 finish in ONE launch: fold the remaining (<= 64) sorted lists, optionally
// re-score the kk survivors in fp32 (rerank), rank by (distance, index), write k.
template <bool RERANK>
__global__ __launch_bounds__(1024) void final_sorted_kernel(const float* __restrict__ in_s, const int* __restrict__ in_i,
                                                            int n_lists, int kk, int k, long long row_offset,
                                                            const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_b16, int ld,
                                                            const float* __restrict__ qn,
                                                            long long* __restrict__ out_idx,
                                                            float* __restrict__ out_dist,
                                                            const int* __restrict__ alt_flag = nullptr,
                                                            const float* __restrict__ alt_s = nullptr,
                                                            const int* __restrict__ alt_i = nullptr, int alt_lists = 0) {
  __shared__ float ls[8 * 64];
  __shared__ int li[8 * 64];
  const int q = blockIdx.x;
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
  if (alt_flag && *alt_flag) { in_s = alt_s; in_i = alt_i; n_lists = alt_lists; }   // block-uniform
  WaveTopK<1> tk;
  tk.init(kk);
  fold_sorted_lists(tk, in_s + (size_t)q * n_lists * kk, in_i + (size_t)q * n_lists * kk, w, n_lists, 16, kk);
  block_tree_merge<16>(tk, ls, li);
  if (w == 0) tk.store(ls, li);
  __syncthreads();
  if constexpr (RERANK) {
    // each wave re-scores candidates w, w+16, w+32, w+48: all row loads in flight together
    const f32x4* qv = reinterpret_cast<const f32x4*>(qn + (size_t)q * ld);
    const int c4 = ld >> 2;
    int ids[4];
    float acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int c = w + 16 * t; ids[t] = c < kk ? li[c] : kIdxSentinel; acc[t] = 0.f; }
    for (int j = lane; j < c4; j += 64) {
      const f32x4 b = qv[j];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (ids[t] != kIdxSentinel) {  // wave-uniform
          const f32x4 a = stored_row4(rows_f32, rows_b16, ids[t], ld, j);
          acc[t] = fmaf(a.x, b.x, acc[t]); acc[t] = fmaf(a.y, b.y, acc[t]);
          acc[t] = fmaf(a.z, b.z, acc[t]); acc[t] = fmaf(a.w, b.w, acc[t]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = w + 16 * t;
      if (ids[t] != kIdxSentinel) {
        const float tot = wave_sum(acc[t]);
        if (lane == 0) ls[c] = tot;
      }
    }
    __syncthreads();
  }
  rank_and_write(ls, li, kk, k, row_offset, out_idx + (size_t)q * k, out_dist + (size_t)q * k);
}

// Cross-shard merge after the all-gather: g sorted lists per query, layout
// [g][b][k]; rank = own position + binary-searched counts in the other lists.
__global__ __launch_bounds__(256) void merge_shards_kernel(const float* __restrict__ cd, long long dstride,
                                                           const long long* __restrict__ ci, long long istride,
                                                           int g, int k,
                                                           long long* __restrict__ out_idx,
                                                           float* __restrict__ out_dist) {
  const int q = blockIdx.x;
  for (int t = threadIdx.x; t < k; t += blockDim.x) { out_idx[(size_t)q * k + t] = -1; out_dist[(size_t)q * k + t] = INFINITY; }
  __syncthreads();
  for (int t = threadIdx.x; t < g * k; t += blockDim.x) {
    const int a = t / k, p = t - a * k;
    const long long id = ci[(size_t)a * istride + (size_t)q * k + p];
    if (id < 0) continue;
    const float d = cd[(size_t)a * dstride + (size_t)q * k + p];
    int rank = p;
    for (int s = 0; s < g; ++s) {
      if (s == a) continue;
      const float* od = cd + (size_t)s * dstride + (size_t)q * k;
      const long long* oi = ci + (size_t)s * istride + (size_t)q * k;
      int lo = 0, hi = k;  // count entries of list s that are better than (d,id)
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const long long mi = oi[mid];
        const bool mb = (mi >= 0) && dist_better(od[mid], mi, d, id);
        if (mb) lo = mid + 1; else hi = mid;
      }
      rank += lo;
    }
    if (rank < k) { out_idx[(size_t)q * k + rank] = id; out_dist[(size_t)q * k + rank] = d; }
  }
}

// ---- add path -------------------------------------------------------------
// L2-normalise rows (fp64 sum, 1/(sqrt+1e-30) as cosine space prescribes) and
// store fp32 and/or bf16 copies with leading dimension ld (zero padded).
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, int in_ld, long long n,
                                                             int dim, int ld, float* __restrict__ out_f32,
                                                             uint16_t* __restrict__ out_bf16) {
  const int lane = lane_id();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = w; r < n; r += nw) {
    const float* x = in + r * in_ld;
    double ss = 0.0;
    for (int c = lane; c < dim; c += 64) { const double v = (double)x[c]; ss = fma(v, v, ss); }   // scan_fused.h's prologue does the same
    ss = wave_sum_f64(ss);
    const double inv = 1.0 / (sqrt(ss) + 1e-30);
    for (int c = lane; c < ld; c += 64) {
      const float y = c < dim ? (float)((double)x[c] * inv) : 0.f;
      if (out_f32) out_f32[r * ld + c] = y;
      if (out_bf16) out_bf16[r * ld + c] = f32_to_bf16_rne(y);
    }
  }
}

// The queries of a batched (MFMA-path) search in ONE launch: rows < b are normalised exactly as normalize_rows_kernel does
// (fp32 copy for the rerank + its bf16 rounding for the pass kernels), rows [b, b_pad) of the bf16 image are zero, and
// the batch's overflow flag is cleared — normalize_rows_kernel + queries_to_bf16_kernel were two launches and a gap
// (~10 us of a 0.37 ms step at an 8-GPU shard's size).
__global__ __launch_bounds__(256) void prep_queries_bf16_kernel(const float* __restrict__ in, int dim, int b, int b_pad,
                                                                float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16,
                                                                int* __restrict__ clear_flag) {
  if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;
  const int lane = lane_id();
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (blockDim.x >> 6);
  for (int r = w; r < b_pad; r += nw) {
    if (r >= b) {   // wave-uniform
      for (int c = lane; c < dim; c += 64) out_bf16[(size_t)r * dim + c] = 0;
      continue;
    }
    const float* x = in + (size_t)r * dim;
    double ss = 0.0;
    for (int c = lane; c < dim; c += 64) { const double v = (double)x[c]; ss = fma(v, v, ss); }
    ss = wave_sum_f64(ss);
    const double inv = 1.0 / (sqrt(ss) + 1e-30);
    for (int c = lane; c < dim; c += 64) {
      const float y = (float)((double)x[c] * inv);
      out_f32[(size_t)r * dim + c] = y;
      out_bf16[(size_t)r * dim + c] = f32_to_bf16_rne(y);
    }
  }
}

// The same for the add path's common case (dim a multiple of 4, <= 2048, 16-byte aligned input rows): a wave brings its
// whole row into registers with NV float4 loads per lane (all in flight together), squares and sums in fp64 as above, and
// writes float4 / 4 x bf16 per lane.  One read of the input instead of two, 16-byte accesses instead of 4- and 2-byte
// ones: HBM-bound copy work (read 4, write 4 [+2] bytes per element).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int NV>
__global__ __launch_bounds__(256) void normalize_rows_vec_kernel(const float* __restrict__ in, int in_ld, long long n,
                                                                 int dim, int ld, float* __restrict__ out_f32,
                                                                 uint16_t* __restrict__ out_bf16) {
  const int lane = lane_id();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  const int nv_in = dim >> 2, nv_out = ld >> 2;
  // two rows per wave and step (round 3): their 2 x NV loads are in flight together, and the stores are non-temporal —
  // the rows are not read again before the scan streams them, so they need not displace anything in L2
  constexpr int RW = NV <= 4 ? 2 : 1;
  for (long long r0 = w * RW; r0 < n; r0 += nw * RW) {
    f32x4 v[RW][NV];
#pragma unroll
    for (int u = 0; u < RW; ++u) {
      const long long r = r0 + u < n ? r0 + u : n - 1;   // (a clamped second row is loaded again, never stored)
      const f32x4* x = reinterpret_cast<const f32x4*>(in + r * in_ld);
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int j = lane + 64 * t;
        v[u][t] = j < nv_in ? __builtin_nontemporal_load(x + j) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) {
      const long long r = r0 + u;
      if (r >= n) break;   // wave-uniform
      double ss = 0.0;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        ss += (double)v[u][t].x * (double)v[u][t].x; ss += (double)v[u][t].y * (double)v[u][t].y;
        ss += (double)v[u][t].z * (double)v[u][t].z; ss += (double)v[u][t].w * (double)v[u][t].w;
      }
      ss = wave_sum_f64(ss);
      const double inv = 1.0 / (sqrt(ss) + 1e-30);
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const int j = lane + 64 * t;
        if (j < nv_out) {
          f32x4 y;
          y.x = (float)((double)v[u][t].x * inv); y.y = (float)((double)v[u][t].y * inv);
          y.z = (float)((double)v[u][t].z * inv); y.w = (float)((double)v[u][t].w * inv);
          if (out_f32) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(out_f32 + r * ld) + j);
          if (out_bf16) {
            u32x2 p;
            p.x = (uint32_t)f32_to_bf16_rne(y.x) | ((uint32_t)f32_to_bf16_rne(y.y) << 16);
            p.y = (uint32_t)f32_to_bf16_rne(y.z) | ((uint32_t)f32_to_bf16_rne(y.w) << 16);
            __builtin_nontemporal_store(p, reinterpret_cast<u32x2*>(out_bf16 + r * ld) + j);
          }
        }
      }
    }
  }
}

// copy stored rows (already normalised) into a query buffer as fp32
__global__ void rows_to_f32_kernel(const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_bf16,
                                   int ld, long long row0, int n, float* __restrict__ out) {
  const long long total = (long long)n * ld;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long src = row0 * ld + i;
    out[i] = rows_f32 ? rows_f32[src] : __uint_as_float((uint32_t)rows_bf16[src] << 16);
  }
}

// gather rows by index -> dense fp32 [n][dim]
__global__ void gather_rows_kernel(const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_bf16,
                                   int ld, int dim, const long long* __restrict__ idx, long long n,
                                   float* __restrict__ out) {
  const int lane = lane_id();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = w; r < n; r += nw) {
    const long long src = idx[r];
    for (int c = lane; c < dim; c += 64)
      out[r * dim + c] = rows_f32 ? rows_f32[src * ld + c] : __uint_as_float((uint32_t)rows_bf16[src * ld + c] << 16);
  }
}

// rows of a search result: idx holds GLOBAL indices (local + row_offset) or -1 -> dense fp32 [n][dim], zeros behind -1
__global__ void gather_result_rows_kernel(const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_bf16,
                                          int ld, int dim, const long long* __restrict__ idx, long long n,
                                          long long row_offset, float* __restrict__ out) {
  const int lane = lane_id();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = w; r < n; r += nw) {
    const long long g = idx[r];
    const long long src = g - row_offset;
    for (int c = lane; c < dim; c += 64)
      out[r * dim + c] = g < 0 ? 0.f : rows_f32 ? rows_f32[src * ld + c] : __uint_as_float((uint32_t)rows_bf16[src * ld + c] << 16);
  }
}

__global__ void fill_empty_kernel(long long* __restrict__ out_idx, float* __restrict__ out_dist, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out_idx[i] = -1; out_dist[i] = INFINITY; }
}

// load path: rows on file are already unit vectors -> copy verbatim (+ derive bf16), pad to ld
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ in, long long n, int dim, int ld,
                                                        float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16) {
  const int lane = lane_id();
  const long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long r = w; r < n; r += nw)
    for (int c = lane; c < ld; c += 64) {
      const float y = c < dim ? in[r * dim + c] : 0.f;
      if (out_f32) out_f32[r * ld + c] = y;
      if (out_bf16) out_bf16[r * ld + c] = f32_to_bf16_rne(y);
    }
}

}  // namespace vdbk
