// synth.h — counter-based synthetic row generator, bit-identical to the CPU
// oracle's (oracle/vdb_oracle.c) so corpora are generated in HBM and any row
// can be regenerated on the host.  Bench/test utility only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vdb.h"

namespace vdbk {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rowkey(uint64_t seed, uint64_t row) {
  return mix64(seed ^ (row * 0xD6E8FEB86659FD93ULL));
}
__device__ __forceinline__ float uval(uint64_t key, uint32_t col) {
  const uint64_t h = mix64(key + (uint64_t)col);
  const int32_t t = (int32_t)(h >> 40);
  return (float)(t - 8388608) * (1.0f / 8388608.0f);
}

struct SynthRowCtx {  // per-row constants
  uint64_t key;       // noise / iso key
  uint64_t cen_key;   // centroid key (kind 1) or base noise key (kind 2)
  uint64_t base_cen_key;
  float noise, base_noise;
  int kind, base_kind;
};

__device__ __forceinline__ SynthRowCtx synth_row_ctx(const vdb_synth_spec& s, int64_t row) {
  SynthRowCtx c;
  c.kind = s.kind;
  c.noise = s.noise;
  c.key = rowkey(s.seed, (uint64_t)row);
  c.cen_key = 0; c.base_cen_key = 0; c.base_noise = 0.f; c.base_kind = 0;
  if (s.kind == 1) {
    const int64_t cl = row % (int64_t)(s.n_clusters > 0 ? s.n_clusters : 1);
    c.cen_key = rowkey(s.seed ^ 0xC1A55E5ULL, (uint64_t)cl);
  } else if (s.kind == 2) {
    const int64_t src = (int64_t)(mix64(c.key ^ 0x51C0FFEEULL) % (uint64_t)(s.base_n > 0 ? s.base_n : 1));
    c.base_kind = s.base_kind;
    c.base_noise = s.base_noise;
    c.cen_key = rowkey(s.base_seed, (uint64_t)src);
    if (s.base_kind == 1) {
      const int64_t cl = src % (int64_t)(s.base_n_clusters > 0 ? s.base_n_clusters : 1);
      c.base_cen_key = rowkey(s.base_seed ^ 0xC1A55E5ULL, (uint64_t)cl);
    }
  }
  return c;
}

__device__ __forceinline__ float synth_value(const SynthRowCtx& c, uint32_t col) {
  const float u = uval(c.key, col);
  if (c.kind == 0) return u;
  if (c.kind == 1) return __fmaf_rn(c.noise, u, uval(c.cen_key, col));
  float base = uval(c.cen_key, col);
  if (c.base_kind == 1) base = __fmaf_rn(c.base_noise, base, uval(c.base_cen_key, col));
  return __fmaf_rn(c.noise, u, base);
}

// one wave per row
__global__ void synth_rows_kernel(float* __restrict__ out, int64_t row0, int64_t n, int dim, int ld,
                                  vdb_synth_spec spec) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = w; r < n; r += nw) {
    const SynthRowCtx c = synth_row_ctx(spec, row0 + r);
    for (int col = lane; col < ld; col += 64) out[r * ld + col] = col < dim ? synth_value(c, (uint32_t)col) : 0.f;
  }
}

}  // namespace vdbk
