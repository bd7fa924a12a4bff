// scan_stream.h — pass kernel of the batched bf16 path for SMALL batches (3..64 queries): HBM-bound, one corpus pass serves
// the whole batch.
//
// Why: a batch of up to 64 queries is as HBM-bound as one query (intensity <= 64 flop/B against a ridge of 312), but the
// GEMV scan (scan_gemv.h) pays FMAs and a wave reduction per query (4 queries per pass cost 1.64x one), and the tiled MFMA
// kernels (scan_mfma6.h) stage the corpus through block-shared LDS tiles for a reuse one query tile does not have.
//   * a wave owns 16 corpus rows at a time and multiplies them with v_mfma_f32_16x16x32_bf16 (A = 16 rows x 32 K) against
//     NQ blocks of 16 queries;
//   * the (<= 64) queries live in LDS for the whole kernel as the B operand, rows padded to 1552 bytes so that the 16
//     lanes of a ds_read_b128 phase hit 16 different bank groups; an opaque offset keeps hipcc from hoisting the
//     loop-invariant fragment reads into 96 NQ registers;
//   * the corpus comes in WHOLE cache lines: a wave brings its 16 rows in slabs of 16 x 128 bytes = two coalesced LDS-DMA
//     instructions (global_load_lds_dwordx4: 8 lanes per row-line, 8 rows per instruction, no VGPR round trip) into a
//     PRIVATE ring of slabs and reads the A fragments back in MFMA layout with ds_read_b128.  LDS position (16-byte units)
//     of (row m, chunk ch) inside a slab: m*8 + (ch ^ ((m>>1)&7)) — the swizzle is applied on the SOURCE address of each DMA
//     lane and makes the 16 lanes of a fragment read fall into 16 different bank groups.  No barrier after the query image
//     is in place: a wave only ever reads what its own DMA wrote, behind a counted s_waitcnt vmcnt (the younger slabs of
//     the ring stay in flight);
//   * the epilogue is the protocol's (scan_mfma3.h): MODE 0 writes the dense sample's scores, MODE 1 appends rows scoring
//     >= the query's threshold to its candidate buffer.  A lane holds 4 scores of one query per 16-query block: an integer
//     max tree + one ballot decide the common (no hit) case; hits wait in registers and reserve their slots together.
// The work unit is a 16-row slice of one of the protocol's 256-row tiles, dealt round-robin to the waves of the grid, so
// that the chip walks the corpus as one contiguous window (like the GEMV scan).
// Algorithmic bytes per final pass = n_rows * 1536: the kernel is priced against the HBM roofline (6.6-6.7 TB/s measured).
//
// History: the first form fed the MFMAs straight from global_load_dwordx4 (lane l = row l%16, K-chunk l/16: exactly the A
// fragment, no LDS for the corpus at all).  Its 64-byte row pieces are HALF cache lines, which the memory path delivers
// ~13 % slower: 2.66-2.73 ms per step at 10M rows against 2.36-2.39 for this one (profiles/r02_batch_sweep_forced.txt); it
// was retired once this form also served the dense pass.
#pragma once
#include "scan_mfma3.h"

namespace vdbk {

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kStMaxB = 64;                      // queries this kernel serves at most
constexpr int kStThreads = 512;                  // 8 waves share one LDS query image
constexpr int kStWaves = kStThreads / 64;
constexpr int kStQStride = kMfRowBytes + 16;     // LDS row stride of a query: 1552 B (1536 + one 16-byte bank-group shift)
constexpr int kStRows = 16;                      // corpus rows per wave tile (M of the MFMA)
constexpr int kStSub = kMfBM / kStRows;          // wave tiles per 256-row protocol tile

__host__ __device__ constexpr size_t st_smem_bytes(int nq) { return (size_t)nq * 16 * kStQStride; }

constexpr int kStSlabBytes = kStRows * 128;          // 2 KiB: 16 rows x one 128-byte line = two k32 steps
constexpr int kStSlabsPerTile = kMfRowBytes / 128;   // 12
// ring depth per wave: 6 slabs (12 KiB, 10 KiB in flight) beside <= 32 queries; 3 beside 33..64 (their image takes 97 KiB)
#ifndef VDB_ST_RING
#define VDB_ST_RING 6
#endif
__host__ __device__ constexpr int st_ring_slabs(int nq) { return nq <= 2 ? VDB_ST_RING : 3; }
__host__ __device__ constexpr size_t st_lds_smem_bytes(int nq) { return st_smem_bytes(nq) + (size_t)kStWaves * st_ring_slabs(nq) * kStSlabBytes; }

template <bool NT>
__device__ __forceinline__ void st_glds16(const unsigned char* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, NT ? 2 : 0);
}

template <int MODE, int NQ, bool NT>
__global__ __launch_bounds__(kStThreads, 2) void scan_stream_lds_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b,
    int n_pass_tiles, int tile_stride, int skip_n, int self0, const float* __restrict__ thr_in,
    float* __restrict__ dense, int* __restrict__ cand_cnt, float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int xcd_slices) {
  extern __shared__ __align__(16) unsigned char smem[];   // [query image][8 waves x kStSlabs slabs]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15;
  const int c = lane >> 4;

  for (int i = tid; i < NQ * 16 * (kMfRowBytes / 16); i += kStThreads) {
    const int qr = i / (kMfRowBytes / 16), ch = i - qr * (kMfRowBytes / 16);
    *reinterpret_cast<st_u32x4*>(smem + qr * kStQStride + ch * 16) =
        *reinterpret_cast<const st_u32x4*>(qb16 + (size_t)qr * kMfRowBytes + ch * 16);
  }
  __syncthreads();   // the only barrier

  float thr[NQ];
  int thr_bits[NQ];
#pragma unroll
  for (int nq = 0; nq < NQ; ++nq) {
    thr[nq] = (MODE == 1 && thr_in) ? thr_in[min(nq * 16 + m, b - 1)] : -INFINITY;
    thr_bits[nq] = thr[nq] > 0.f ? __float_as_int(thr[nq]) : (int)0x80000000;
  }

  const long long n_units = (long long)n_pass_tiles * kStSub;
  const long long nw = (long long)gridDim.x * kStWaves;
  // blocks b, b + 8, ... share an XCD: with xcd_slices they take ADJACENT units, so that each grid-wide step is eight
  // contiguous per-XCD slices of one window (the read probe: 7.15-7.19 TB/s against 7.06-7.09 for block-by-block
  // round-robin, profiles/r03_placement.txt)
  const int blk = (xcd_slices && (gridDim.x & 7u) == 0u) ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  long long u = (long long)blk * kStWaves + w;
  if (u >= n_units) return;
  auto unit_row0 = [&](long long uu) -> int {
    const int t = (int)(uu / kStSub), sub = (int)(uu - (long long)t * kStSub);
    return pass_tile(t, tile_stride, skip_n) * kMfBM + sub * kStRows;
  };
  constexpr int kStSlabs = st_ring_slabs(NQ);
  static_assert(kStSlabsPerTile % kStSlabs == 0, "ring slots must be compile-time");
  unsigned char* ring = smem + st_smem_bytes(NQ) + w * (kStSlabs * kStSlabBytes);
  // DMA source offsets of this lane inside a tile, for the slab's two instructions (rows 0-7, rows 8-15)
  unsigned src_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mr = 8 * i + (lane >> 3), ch = (lane & 7) ^ ((mr >> 1) & 7);
    src_off[i] = (unsigned)(mr * kMfRowBytes + ch * 16);
  }
  // fragment read offsets inside a slab, k32 step t of the slab
  int rd[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) rd[t] = (m * 8 + ((4 * t + c) ^ ((m >> 1) & 7))) * 16;
  const unsigned char* qfrag = smem + m * kStQStride + c * 16;

  int row0 = unit_row0(u);
  const unsigned char* pc = rows + (size_t)row0 * kMfRowBytes;   // wave-uniform tile base
  auto issue_slab = [&](const unsigned char* tile, int seg, int slot) {
#pragma unroll
    for (int i = 0; i < 2; ++i) st_glds16<NT>(tile + seg * 128 + src_off[i], ring + slot * kStSlabBytes + i * 1024);
  };
#pragma unroll
  for (int j = 0; j < kStSlabs; ++j) issue_slab(pc, j, j);

  constexpr int kStPend = 2;
  float pend_v[kStPend];
  int pend_row[kStPend], pend_q[kStPend];
  int pend_n = 0;
#pragma unroll
  for (int sl = 0; sl < kStPend; ++sl) { pend_v[sl] = 0.f; pend_row[sl] = 0; pend_q[sl] = 0; }
  auto flush_pending = [&]() {
    int pos[kStPend];
#pragma unroll
    for (int sl = 0; sl < kStPend; ++sl) pos[sl] = sl < pend_n ? atomicAdd(&cand_cnt[pend_q[sl] * kMfCntStride], 1) : cap;   // all in flight, one wait
    // the atomics may return out of order with the DMA loads: drain, so that the counted waits of the next tile count loads only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int sl = 0; sl < kStPend; ++sl)
      if (pos[sl] < cap) { cand_s[(size_t)pend_q[sl] * cap + pos[sl]] = pend_v[sl]; cand_i[(size_t)pend_q[sl] * cap + pos[sl]] = pend_row[sl]; }
    pend_n = 0;
  };

  for (;;) {
    const long long un = u + nw;
    const bool has_next = un < n_units;
    const int row0n = has_next ? unit_row0(un) : row0;   // the last unit refills from itself: harmless, never consumed
    const unsigned char* pn = rows + (size_t)row0n * kMfRowBytes;
    f32x4_t acc[NQ];
#pragma unroll
    for (int nq = 0; nq < NQ; ++nq) acc[nq] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    int qoff = 0;
    asm volatile("" : "+v"(qoff));   // keeps the (loop-invariant) query fragment reads inside the loop, see above
#pragma unroll
    for (int j = 0; j < kStSlabsPerTile; ++j) {
      constexpr int kYounger = 2 * (kStSlabs - 1);   // DMA instructions of the younger slabs still allowed in flight
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kYounger) : "memory");   // slab j has landed
      const unsigned char* slab = ring + (j % kStSlabs) * kStSlabBytes;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(slab + rd[t]);
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
          const bf16x8_t q = *reinterpret_cast<const bf16x8_t*>(qfrag + qoff + nq * 16 * kStQStride + (2 * j + t) * 64);
          acc[nq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, q, acc[nq], 0, 0, 0);
        }
      }
      // the fragments of this slab are in registers (the MFMAs above waited for them): its slot can be refilled
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int jn = j + kStSlabs;
      if (jn < kStSlabsPerTile) issue_slab(pc, jn, j % kStSlabs); else issue_slab(pn, jn - kStSlabsPerTile, j % kStSlabs);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: as in scan_stream_kernel ----
    const int t_local = (int)(u / kStSub), sub = (int)(u - (long long)t_local * kStSub);
    if (MODE == 0) {
#pragma unroll
      for (int nq = 0; nq < NQ; ++nq) {
        const int qg = nq * 16 + m;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int rloc = sub * kStRows + 4 * c + reg;
          const int row = row0 + 4 * c + reg;
          const bool ok = (row < n_rows) && !(self0 >= 0 && row == self0 + qg);
          if (qg < b) dense[(size_t)qg * kMf3DenseLd + t_local * kMfBM + rloc] = ok ? acc[nq][reg] : -INFINITY;
        }
      }
    } else {
      bool any = false;
#pragma unroll
      for (int nq = 0; nq < NQ; ++nq) {
        const int mx = max(max(__float_as_int(acc[nq][0]), __float_as_int(acc[nq][1])),
                           max(__float_as_int(acc[nq][2]), __float_as_int(acc[nq][3])));
        any = any || (mx >= thr_bits[nq]);
      }
      if (__ballot(any) != 0ull) {
        // A hit waits in its lane's registers (kStPend places) until some lane needs one more; then every waiting hit
        // reserves its slot at once (scan_mfma5.h does the same).  The returning atomics share the counter with the DMA
        // loads, so a flush drains the wave's load ring: once per ~a dozen hits this way instead of once per tile with a hit.
#pragma unroll
        for (int nq = 0; nq < NQ; ++nq) {
          const int qg = nq * 16 + m;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = row0 + 4 * c + reg;
            const bool pass = (acc[nq][reg] >= thr[nq]) && (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
            if (__ballot(pass) == 0ull) continue;
            if (__ballot(pass && pend_n == kStPend) != 0ull) flush_pending();
            if (pass) {
#pragma unroll
              for (int sl = 0; sl < kStPend; ++sl)
                if (pend_n == sl) { pend_v[sl] = acc[nq][reg]; pend_row[sl] = row; pend_q[sl] = qg; }
              ++pend_n;
            }
          }
        }
        if (self0 == -2) flush_pending();   // diagnostic (mfma_ablate bit 128): one flush per tile with a hit, as before the waiting places
      }
    }
    if (!has_next) break;
    u = un; row0 = row0n; pc = pn;
  }
  if (MODE == 1) flush_pending();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing refills must land before the LDS is handed on
}

// ---- host side --------------------------------------------------------------------------------------
inline int st_blocks(int b) { const int nq = (b + 15) / 16; return nq <= 1 ? 1 : nq <= 2 ? 2 : 4; }

inline hipError_t stream_prepare_device() {
  hipError_t e;
#define ST_ATTR(MODE, NQ, NT) \
  if ((e = hipFuncSetAttribute((const void*)scan_stream_lds_kernel<MODE, NQ, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)st_lds_smem_bytes(NQ))) != hipSuccess) return e;
  ST_ATTR(0, 1, false) ST_ATTR(0, 2, false) ST_ATTR(0, 4, false)
  ST_ATTR(1, 1, false) ST_ATTR(1, 1, true) ST_ATTR(1, 2, false) ST_ATTR(1, 2, true) ST_ATTR(1, 4, false) ST_ATTR(1, 4, true)
#undef ST_ATTR
  return hipSuccess;
}

// one pass over n_pass_tiles protocol tiles (pass_tile(i, stride, skip_n)); mode 0 = dense sample, 1 = append
inline void stream_launch_pass(int mode, const unsigned char* rows, int n_rows, const unsigned char* qb16, int b, int n_pass_tiles,
                               int tile_stride, int skip_n, int self0, const float* thr, float* dense, int* cand_cnt,
                               float* cand_s, int* cand_i, int cap, int n_cu, int nt, hipStream_t st) {
  const int nq = st_blocks(b);
  const long long units = (long long)n_pass_tiles * kStSub;
  const int grid = (int)std::max<long long>(1, std::min<long long>((long long)n_cu, (units + kStWaves - 1) / kStWaves));   // one block per CU
#define STL_LAUNCH(MODE, NQ, NT)                                                                                                       \
  hipLaunchKernelGGL((scan_stream_lds_kernel<MODE, NQ, NT>), dim3(grid), dim3(kStThreads), st_lds_smem_bytes(NQ), st, rows, n_rows, qb16, b, \
                     n_pass_tiles, tile_stride, skip_n, self0, thr, dense, cand_cnt, cand_s, cand_i, cap, xcd_slices)
  const int xcd_slices = (nt >> 1) & 1;   // nt: bit 0 = non-temporal corpus loads, bit 1 = per-XCD slices
  nt &= 1;
  if (mode == 0) {
    if (nq == 1) STL_LAUNCH(0, 1, false); else if (nq == 2) STL_LAUNCH(0, 2, false); else STL_LAUNCH(0, 4, false);
  } else if (nt) {
    if (nq == 1) STL_LAUNCH(1, 1, true); else if (nq == 2) STL_LAUNCH(1, 2, true); else STL_LAUNCH(1, 4, true);
  } else {
    if (nq == 1) STL_LAUNCH(1, 1, false); else if (nq == 2) STL_LAUNCH(1, 2, false); else STL_LAUNCH(1, 4, false);
  }
#undef STL_LAUNCH
}

}  // namespace vdbk
