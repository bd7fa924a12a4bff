// probes.h — what THIS device delivers on the two ceilings the rooflines are quoted against
// (report against the vendor peaks AND against measured ones): a streaming
// HBM read of the index's own rows and a bare bf16 MFMA loop.  Read through vdb_get_stat.
#pragma once
#include "scan_mfma.h"

namespace vdbk {

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));

// every lane streams 16-byte chunks, U in flight, non-temporal (the scan kernel's access shape)
template <int U>
__global__ __launch_bounds__(256) void probe_read_kernel(const v4u_t* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  unsigned acc = 0;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    v4u_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { const v4u_t a = __builtin_nontemporal_load(p + i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (acc == 0x9E3779B9u) *sink = acc;   // keeps the loads alive; practically never true
}

// operands in registers, 8 independent accumulators per wave, two waves per SIMD
__global__ __launch_bounds__(512) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  bf16x8_t a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (float)((lane + i) & 15)); b[i] = (__bf16)(0.002f * (float)((lane * 3 + i) & 7)); }
  f32x16_t acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][15];
  if (s == 12345.678f) *sink = s;
}

}  // namespace vdbk
