// probes.h — what THIS device delivers on the two ceilings the rooflines are quoted against
// (report against the vendor peaks AND against measured ones): a streaming
// HBM read of the index's own rows and a bare bf16 MFMA loop.  Read through vdb_get_stat.
#pragma once
#include <hip/hip_runtime.h>

namespace vdbk {

typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
typedef __bf16 pr_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float pr_f32x16_t __attribute__((ext_vector_type(16)));

// The scan kernel's own access shape (scan_gemv.h) with the arithmetic taken out: a wave owns one contiguous
// U KiB tile at a time (U 16-byte loads per lane, 1 KiB per wave-instruction, non-temporal), tiles are dealt
// round-robin to the waves of the grid.  Launched like the scan (256 threads, 2 or 4 blocks per CU), so that
// what it measures is a ceiling FOR the scan, not another kernel's rate.
template <int U>
__global__ __launch_bounds__(256) void probe_read_kernel(const v4u_t* __restrict__ p, size_t n16, unsigned* __restrict__ sink) {
  const size_t lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  const size_t n_tiles = n16 / (U * 64);
  unsigned acc = 0;
  // the next tile's loads go out before this tile's are consumed (two tiles in flight per wave): a bare read that waited
  // for each tile before asking for the next measured BELOW the scan, whose arithmetic overlaps its loads by itself
  v4u_t cur[U], nxt[U];
  size_t t = gw;
  if (t < n_tiles) {
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(p + t * (U * 64) + lane + u * 64);
  }
  for (; t < n_tiles; t += nw) {
    const bool more = t + nw < n_tiles;   // wave-uniform
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = __builtin_nontemporal_load(p + (t + nw) * (U * 64) + lane + u * 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  if (acc == 0x9E3779B9u) *sink = acc;   // keeps the loads alive; practically never true
}

// The same bare read in the FUSED scan's launch shape (scan_fused.h): block b's waves take tiles t_begin + wid, + nwv, ...
// of the range its block group was dealt — the partition the scan's self-tuned per-XCD shares give, computed on the
// host from the same share table — so that the figure stays a ceiling for a scan that balances its XCDs.
// A block's place in the fused scan's tile partition (TileWalk, scan_gemv.h): member s of its class's sequence lies at
// base + (s / q) * R + s % q while s < full_q, at tail_base + (s - full_q) behind that; the block's four waves are
// members wid0 ... wid0 + 3 of every nwv.
struct ProbeRange { int wid0, nwv, n_seq, base, q, R, full_q, tail_base; };
// PACE: units of 64 cycles a wave idles between consuming a tile and asking for the next one.  The scan spends ~300-400
// cycles of arithmetic there, which spaces its waves' requests out; a bare read that re-issues at once keeps all waves in
// step with the memory latency and measured 2 % BELOW the scan.  With the pause the probe is a read in the scan's own
// rhythm, and the best over a few pauses is the ceiling the scan is held against.
template <int U, int PACE>
__global__ __launch_bounds__(256) void probe_read_ranges_kernel(const v4u_t* __restrict__ p, const ProbeRange* __restrict__ rg, unsigned* __restrict__ sink) {
  const size_t lane = threadIdx.x & 63;
  const ProbeRange r = rg[blockIdx.x];
  unsigned acc = 0;
  v4u_t cur[U], nxt[U];
  auto tile_of = [&](int s) -> size_t {
    return s < r.full_q ? (size_t)r.base + (size_t)(s / r.q) * (size_t)r.R + (size_t)(s % r.q) : (size_t)r.tail_base + (size_t)(s - r.full_q);
  };
  int s = r.wid0 + (int)(threadIdx.x >> 6);
  if (s < r.n_seq) {
    const size_t t = tile_of(s);
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = __builtin_nontemporal_load(p + t * (U * 64) + lane + u * 64);
  }
  for (; s < r.n_seq; s += r.nwv) {
    const bool more = s + r.nwv < r.n_seq;
    if (more) {
      const size_t t = tile_of(s + r.nwv);
#pragma unroll
      for (int u = 0; u < U; ++u) nxt[u] = __builtin_nontemporal_load(p + t * (U * 64) + lane + u * 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= cur[u].x ^ cur[u].y ^ cur[u].z ^ cur[u].w;
    if (PACE > 0) __builtin_amdgcn_s_sleep(PACE);
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
  }
  if (acc == 0x9E3779B9u) *sink = acc;
}

// Operands of the MFMA probes: RANDOM values (a unit vector's elements are ~ +-0.036), two sets used in turn, so
// that the multiplier inputs toggle as they do on real rows — on constant or zero operands the chip holds a higher
// clock and the probe overstates what a real kernel can reach.
__device__ __forceinline__ void probe_random_operands(pr_bf16x8_t (&a)[2], pr_bf16x8_t (&b)[2], int lane) {
  unsigned x = 0x9E3779B9u * (unsigned)(lane + 1) + blockIdx.x * 0x85EBCA6Bu + threadIdx.x;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      a[s][i] = (__bf16)(((float)(int)(x & 0xffff) - 32768.f) * (0.07f / 32768.f));
      x ^= x << 13; x ^= x >> 17; x ^= x << 5;
      b[s][i] = (__bf16)(((float)(int)(x & 0xffff) - 32768.f) * (0.07f / 32768.f));
    }
}

// operands in registers, 8 independent accumulators per wave, two waves per SIMD
__global__ __launch_bounds__(512) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  pr_bf16x8_t a[2], b[2];
  probe_random_operands(a, b, lane);
  pr_f32x16_t acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 1], b[(j >> 1) & 1], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][15];
  if (s == 12345.678f) *sink = s;
}

// the same loop on v_mfma_f32_16x16x32_bf16 (same FLOPs per instruction-cycle; the guide reports a higher clock under it)
typedef float pr_f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void probe_mfma16_kernel(int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  pr_bf16x8_t a[2], b[2];
  probe_random_operands(a, b, lane);
  pr_f32x4_t acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = pr_f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 1], b[(j >> 1) & 1], acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][3];
  if (s == 12345.678f) *sink = s;
}

}  // namespace vdbk
