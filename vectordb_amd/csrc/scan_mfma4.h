// scan_mfma4.h — warp-specialised form of the batched pass kernel (scan_mfma3.h).
//
// Measurement that motivates it (MI355X, 10M x 768, B=1024): with every wave both
// staging (global_load_lds) and computing behind one block barrier per K-step, loads
// alone ran at 11.6 ms, MFMA alone at ~13 ms, together at ~21 ms — the two never
// overlapped: all 8 waves stall in the same phase, and a wave blocked issuing LDS-DMA
// cannot issue MFMAs.  Here the roles are split, with no block barrier in the loop:
//   waves 0-7  consumers : 4x2 layout, 64 corpus rows x 64 queries each; per K-step they
//                          wait on the stage's FULL counter (LDS), read 16 fragments,
//                          issue 16 MFMAs, bump the stage's FREE counter.
//   waves 8-11 producers : one per SIMD; each stages a quarter of every K-step (12
//                          global_load_lds_dwordx4) into a 3-stage ring, runs up to two
//                          stages ahead, publishes a stage (FULL += 1) once its counted
//                          s_waitcnt vmcnt(12) shows that stage's loads have landed, and
//                          reuses a slot only after all 8 consumers released it.
// Counters only grow (no resets, no ABA); every spin is bounded and a timeout raises the
// batch's overflow flag, which routes it through the exact list kernel (scan_mfma.h).
// Same tile geometry, LDS image, swizzle, epilogues and exactness argument as scan_mfma3.h.
#pragma once
#include "scan_mfma3.h"

namespace vdbk {

constexpr int kWsConsumers = 8;
constexpr int kWsProducers = 4;
constexpr int kWsThreads = (kWsConsumers + kWsProducers) * 64;   // 768: 3 waves per SIMD, <= 168 VGPRs
constexpr int kWsSpinLimit = 1 << 22;

__host__ __device__ constexpr size_t ws_smem_bytes() { return (size_t)kMfNumStages * kMfStage + 64; }

// The ring counters are touched with explicit DS instructions: a compiler-generated LDS access
// while LDS-DMA loads are pending makes hipcc drain them first (s_waitcnt vmcnt(0)), which would
// serialise the producers.  Each asm read waits for its own result (lgkmcnt(0) inside the
// statement), so the compiler's counters stay consistent.
__device__ __forceinline__ unsigned ws_lds_addr(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ int ws_lds_read(unsigned addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void ws_lds_add1(unsigned addr) {
  const int one = 1;
  asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(one) : "memory");
}
// wait until the LDS counter at addr >= target (written by other waves of the block); bounded
__device__ __forceinline__ bool ws_wait_ge(unsigned addr, int target) {
  int n = 0;
  while (__builtin_amdgcn_readfirstlane(ws_lds_read(addr)) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (++n > kWsSpinLimit) return false;
  }
  return true;
}

template <int MODE>
__global__ __launch_bounds__(kWsThreads) void scan_mfma_ws_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int* __restrict__ fail_flag, int ablate, int skip_n) {
  extern __shared__ __align__(16) unsigned char smem[];   // 3 stages + 64 B of counters (one LDS object)
  int* ctr = reinterpret_cast<int*>(smem + kMfNumStages * kMfStage);   // [0..2] FULL, [4..6] FREE
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

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
  const int tile0 = sp * tiles_per_split;
  const int tile1 = min(tile0 + tiles_per_split, n_pass_tiles);
  const int my_tiles = tile1 > tile0 ? tile1 - tile0 : 0;
  const int total_steps = my_tiles * kMfSteps;
  if (total_steps == 0) return;

  if (tid < 16) ctr[tid] = 0;
  __syncthreads();
  const unsigned full_a = ws_lds_addr(ctr), free_a = ws_lds_addr(ctr + 4);   // + 4*slot

  if (w >= kWsConsumers) {
    // ------------------------------ producer ------------------------------
    const int p = w - kWsConsumers;
    // piece j (1 KiB = 8 rows x 128 B) of an operand tile: rows 8j..8j+7; this wave takes pieces p, p+4, ...
    // so (row>>1)&7 = (4p + (lane>>4)) & 7 for all of them: one constant per-lane source offset
    const int lrow = 8 * p + (lane >> 3);
    const unsigned lane_off = (unsigned)(lrow * kMfRowBytes + (((lane & 7) ^ ((4 * p + (lane >> 4)) & 7)) << 4));
    const unsigned char* qblk = qb16 + (size_t)qbase * kMfRowBytes;
    int is_ks = 0, is_slot = 0, is_round = 0, issued = 0;   // cursor of the next stage to issue
    int is_g = pass_tile(tile0, tile_stride, skip_n);                    // its corpus tile (scan_mfma.h)
    auto issue_next = [&]() {
      const unsigned char* abase = rows + (size_t)is_g * (kMfBM * kMfRowBytes) + is_ks * (kMfBK * 2);
      const unsigned char* bbase = qblk + is_ks * (kMfBK * 2);
      unsigned char* st = smem + is_slot * kMfStage + p * 1024;
      if (!(ablate & 1)) {   // ablate bit 0: diagnostic timing runs without the loads (results invalid)
#pragma unroll
        for (int i = 0; i < 8; ++i) mf_glds16(abase + (size_t)i * (32 * kMfRowBytes) + lane_off, st + i * 4096);
#pragma unroll
        for (int i = 0; i < 4; ++i) mf_glds16(bbase + (size_t)i * (32 * kMfRowBytes) + lane_off, st + kMfStageA + i * 4096);
      }
      ++issued;
      if (++is_ks == kMfSteps) { is_ks = 0; is_g = pass_tile_next(is_g, tile_stride, skip_n); }
      if (++is_slot == kMfNumStages) { is_slot = 0; ++is_round; }
    };
    issue_next();
    if (total_steps > 1) issue_next();
    bool ok = true;
    int slot = 0;
    for (int step = 0; step < total_steps && ok; ++step) {
      // stage `step` has landed once only the younger stage's 12 loads are outstanding
      if (!(ablate & 1)) {
        if (issued > step + 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (lane == 0) ws_lds_add1(full_a + 4 * slot);            // publish it
      if (++slot == kMfNumStages) slot = 0;
      if (issued < total_steps) {
        // the slot of stage step+2 held stage step-1: refill it once all consumers released that use
        if (is_round > 0) ok = ws_wait_ge(free_a + 4 * is_slot, kWsConsumers * is_round);
        if (ok) issue_next();
      }
    }
    if (!ok && lane == 0) *fail_flag = 1;
    return;
  }

  // -------------------------------- consumer --------------------------------
  const int wm = w >> 1, wn = w & 1;
  const int hsel = lane >> 5;
  float thr[2] = {-INFINITY, -INFINITY};
  if (MODE == 1 && thr_in) {
    const int q0 = qbase + wn * 64 + (lane & 31);
    thr[0] = thr_in[min(q0, b - 1)];
    thr[1] = thr_in[min(q0 + 32, b - 1)];
  }
  asm volatile("" :: "v"(thr[0]), "v"(thr[1]));   // retire the loads before the loop
  // block filter of the epilogue on the scores' bit patterns (see scan_mfma5.h): exact against a positive
  // threshold; a lane with a non-positive one gets INT_MIN, its filter always fires
  int thr_bits[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) thr_bits[nt] = thr[nt] > 0.f ? __float_as_int(thr[nt]) : (int)0x80000000;
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
  f32x16_t acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  constexpr int kWsPend = 2;
  float pend_v[kWsPend];
  int pend_row[kWsPend], pend_q[kWsPend];
  int pend_n = 0;
#pragma unroll
  for (int sl = 0; sl < kWsPend; ++sl) { pend_v[sl] = 0.f; pend_row[sl] = 0; pend_q[sl] = 0; }
  auto flush_pending = [&]() {
    int pos[kWsPend];
#pragma unroll
    for (int sl = 0; sl < kWsPend; ++sl) pos[sl] = sl < pend_n ? atomicAdd(&cand_cnt[pend_q[sl] * kMfCntStride], 1) : cap;   // all in flight, one wait
#pragma unroll
    for (int sl = 0; sl < kWsPend; ++sl)
      if (pos[sl] < cap) { cand_s[(size_t)pend_q[sl] * cap + pos[sl]] = pend_v[sl]; cand_i[(size_t)pend_q[sl] * cap + pos[sl]] = pend_row[sl]; }
    pend_n = 0;
  };

  int slot = 0, round = 0, ks = 0, tile = 0;
  bool ok = true;
  for (int step = 0; step < total_steps; ++step) {
    if (ok) ok = ws_wait_ge(full_a + 4 * slot, kWsProducers * (round + 1));
    asm volatile("" ::: "memory");
    const unsigned char* sb = smem + slot * kMfStage;
    bf16x8_t a0[4], a1[4], b0[4], b1[4];
    if (!(ablate & 2)) {   // ablate bit 1: diagnostic timing runs without fragment reads / MFMAs
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
    }
    // release the slot: the LDS queue is in order per wave, so this add executes after the reads above
    asm volatile("" ::: "memory");
    if (lane == 0) ws_lds_add1(free_a + 4 * slot);   // no wait: the next step's poll drains lgkmcnt before any counted wait
    if (++slot == kMfNumStages) { slot = 0; ++round; }
    if (++ks < kMfSteps) continue;
    ks = 0;
    if (ablate & 4) { ++tile; continue; }   // ablate bit 2: no epilogue

    // ---------------- epilogue of pass-local tile (tile0 + tile) ----------------
    const long long tile_row0 = (long long)pass_tile(tile0 + tile, tile_stride, skip_n) * kMfBM;
    const bool needs_mask = (tile_row0 + kMfBM > n_rows) || (qbase + kMfBN > b) || (self0 >= 0);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int mt = blk >> 1, nt = blk & 1;
      if (MODE == 1) {   // one integer max tree + one ballot per 32x32 block; hits are ~1e-4 of the scores
        int m = __float_as_int(acc[mt][nt][0]);
#pragma unroll
        for (int reg = 1; reg < 16; ++reg) m = max(m, __float_as_int(acc[mt][nt][reg]));
        if (__ballot(m >= thr_bits[nt]) == 0ull) continue;
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float v = acc[mt][nt][reg];
        const int rloc = wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hsel;
        const int qg = qbase + wn * 64 + nt * 32 + (lane & 31);
        if (MODE == 0) {
          const int row = (int)tile_row0 + rloc;
          const bool okr = (row < n_rows) && !(self0 >= 0 && row == self0 + qg);
          if (qg < b) dense[(size_t)qg * kMf3DenseLd + (tile0 + tile) * kMfBM + rloc] = okr ? v : -INFINITY;
        } else {
          bool pass = v >= thr[nt];
          if (__ballot(pass) == 0ull) continue;
          const int row = (int)tile_row0 + rloc;
          if (needs_mask) pass = pass && (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
          // a hit waits in its lane's registers until some lane needs a third place; then all waiting hits reserve their
          // slots together: one atomic round trip (~1-2 us) per flush instead of one per hit (scan_mfma5.h does the same)
          if (__ballot(pass && pend_n == kWsPend) != 0ull) flush_pending();
          if (pass) {
#pragma unroll
            for (int sl = 0; sl < kWsPend; ++sl)
              if (pend_n == sl) { pend_v[sl] = v; pend_row[sl] = row; pend_q[sl] = qg; }
            ++pend_n;
          }
          if (ablate & 128) flush_pending();   // diagnostic: one round trip per hit, as before the waiting places
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
  if (MODE == 1) flush_pending();
  if (!ok && lane == 0) *fail_flag = 1;
}

}  // namespace vdbk
