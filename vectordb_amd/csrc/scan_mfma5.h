// scan_mfma5.h — wide-tile form of the batched pass kernel: 256 corpus rows x 256 queries
// per block.
//
// Why: the 256x128 kernels (scan_mfma3.h / scan_mfma4.h) stage (1/128 + 1/256) bytes per
// multiply-add column pair through LDS — 184 GB for 10M x 768 against 1024 queries — and the
// L2 -> LDS path delivers ~66 GB/s per CU (~17 TB/s chip-wide, measured here and listed in
// MI355X_MICROARCH.md "Indexed rows: gather into LDS"): 10.5 ms of pure operand delivery,
// as long as the MFMA work itself.  A 256x256 tile stages 123 GB (7 ms), halves the LDS
// fragment reads per MFMA (a wave owns 128 x 64: 6 ds_read_b128 feed 8 MFMAs instead of
// 4 feeding 4), and leaves the matrix pipe as the limiter.
//
// Structure (512 threads = 8 waves, 2(M) x 4(N), two waves per SIMD, <= 256 registers):
//   * two 64 KiB LDS stages [A 256 rows | B 256 rows] x 128 B, chunk-swizzled as before;
//     every wave stages 8 KiB per K-step with 8 global_load_lds_dwordx4;
//   * ONE block barrier per K-step, placed before the last of its four k16 sub-steps:
//     it certifies both that the next stage has landed (each wave waits for its own
//     loads first) and that everybody's reads of the current stage have returned, so the
//     current buffer is refilled for step+2 right behind it — a full step of flight time;
//   * fragments are double-buffered in registers: the reads of sub-step j+1 (across the
//     stage boundary too) are issued before the MFMAs of sub-step j, so no MFMA waits on LDS;
//   * epilogue: per 32x32 block a max-reduction of the lane's 16 scores and one ballot decide
//     whether any score reaches the query's threshold; only then are the 16 examined.
// Thresholds, append buffers, exactness argument and fallback are those of scan_mfma3.h.
#pragma once
#include "scan_mfma3.h"

namespace vdbk {

constexpr int kM5BN = 256;                          // queries per block
constexpr int kM5StageA = kMfBM * kMfBK * 2;        // 32 KiB
constexpr int kM5StageB = kM5BN * kMfBK * 2;        // 32 KiB
constexpr int kM5Stage = kM5StageA + kM5StageB;     // 64 KiB
constexpr int kM5NumStages = 2;
constexpr int kM5Threads = 512;

__host__ __device__ constexpr size_t m5_smem_bytes() { return (size_t)kM5NumStages * kM5Stage; }

template <int MODE>
__global__ __launch_bounds__(kM5Threads) void scan_mfma_wide_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int ablate) {
  extern __shared__ __align__(16) unsigned char smem[];   // the only LDS object: 2 stages
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
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
  const int qbase = qt * kM5BN;
  const int tile0 = sp * tiles_per_split;
  const int tile1 = min(tile0 + tiles_per_split, n_pass_tiles);
  const int my_tiles = tile1 > tile0 ? tile1 - tile0 : 0;
  const int total_steps = my_tiles * kMfSteps;
  if (total_steps == 0) return;

  float thr[2] = {-INFINITY, -INFINITY};
  if (MODE == 1 && thr_in) {
    const int q0 = qbase + wn * 64 + (lane & 31);
    thr[0] = thr_in[min(q0, b - 1)];
    thr[1] = thr_in[min(q0 + 32, b - 1)];
  }
  asm volatile("" :: "v"(thr[0]), "v"(thr[1]));   // retire the loads before the loop (see scan_mfma3.h)

  // staging: LDS position L (16-byte units) of an operand image <- row L>>3, global chunk (L&7)^swz;
  // corpus and query tiles have the same shape, so one set of offsets serves both
  unsigned g_off[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int L = t * kM5Threads + tid, r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
    g_off[t] = (unsigned)(r * kMfRowBytes + c * 16);
  }
  const int wbase16 = w * 1024;
  const unsigned char* qblk = qb16 + (size_t)qbase * kMfRowBytes;

  int is_tile = 0, is_ks = 0;   // staging cursor (scalar)
  auto issue = [&](int buf) {
    const unsigned char* abase = rows + (size_t)(tile0 + is_tile) * tile_stride * (kMfBM * kMfRowBytes) + is_ks * (kMfBK * 2);
    const unsigned char* bbase = qblk + is_ks * (kMfBK * 2);
    unsigned char* st = smem + buf * kM5Stage + wbase16;
    if (!(ablate & 1)) {   // ablate bit 0: diagnostic timing runs without the loads (results invalid)
#pragma unroll
      for (int t = 0; t < 4; ++t) mf_glds16(abase + g_off[t], st + t * (kM5Threads * 16));
#pragma unroll
      for (int t = 0; t < 4; ++t) mf_glds16(bbase + g_off[t], st + kM5StageA + t * (kM5Threads * 16));
    }
    if (++is_ks == kMfSteps) { is_ks = 0; ++is_tile; }
  };

  // fragment reads: k16 sub-step s uses chunk 2s+hsel of rows (lane&31) + 32*block
  const int swz = ((lane & 31) >> 1) & 7;
  const int a_row_off = (wm * 128 + (lane & 31)) * 128;
  const int b_row_off = kM5StageA + (wn * 64 + (lane & 31)) * 128;
  int ck[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ck[s] = ((2 * s + hsel) ^ swz) << 4;

  bf16x8_t fA[2][4], fB[2][2];
  auto read_frags = [&](int fbuf, const unsigned char* sb, int s) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fA[fbuf][mt] = *reinterpret_cast<const bf16x8_t*>(sb + a_row_off + ck[s] + mt * 4096);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) fB[fbuf][nt] = *reinterpret_cast<const bf16x8_t*>(sb + b_row_off + ck[s] + nt * 4096);
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  auto mfma8 = [&](int fbuf) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA[fbuf][mt], fB[fbuf][nt], acc[mt][nt], 0, 0, 0);
  };
  // one k16 sub-step: the next sub-step's 6 fragment reads go out between this one's 8 MFMAs
  auto pin_schedule = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
  };

  // prologue: stage 0 lands, stage 1 goes out, sub-step 0 fragments of stage 0 are read
  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (total_steps > 1) issue(1);
  read_frags(0, smem, 0);

  int buf = 0, ks = 0, tile = 0;
  for (int step = 0; step < total_steps; ++step) {
    const unsigned char* sb = smem + buf * kM5Stage;
    const unsigned char* sbn = smem + (buf ^ 1) * kM5Stage;
    read_frags(1, sb, 1);
    mfma8(0);
    pin_schedule();
    read_frags(0, sb, 2);
    mfma8(1);
    pin_schedule();
    read_frags(1, sb, 3);
    mfma8(0);
    pin_schedule();
    // every read of this stage has returned and my share of the next stage has landed -> barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (step + 2 < total_steps) issue(buf);              // refill the buffer just released, for step+2
    // first fragments of the next stage (after the last step this reads a stale buffer and the values
    // are dropped: keeping it unconditional keeps the wait before the MFMAs a counted one)
    read_frags(0, sbn, 0);
    mfma8(1);
    pin_schedule();
    buf ^= 1;
    if (++ks < kMfSteps) continue;
    ks = 0;
    if (ablate & 4) { ++tile; continue; }   // ablate bit 2: no epilogue

    // ---------------- epilogue of pass-local tile (tile0 + tile) ----------------
    const long long tile_row0 = (long long)(tile0 + tile) * tile_stride * kMfBM;
    const bool needs_mask = (tile_row0 + kMfBM > n_rows) || (qbase + kM5BN > b) || (self0 >= 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int qg = qbase + wn * 64 + nt * 32 + (lane & 31);
        const int rbase = wm * 128 + mt * 32 + 4 * hsel;
        if (MODE == 0) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int rloc = rbase + (reg & 3) + 8 * (reg >> 2);
            const int row = (int)tile_row0 + rloc;
            const bool ok = (row < n_rows) && !(self0 >= 0 && row == self0 + qg);
            if (qg < b) dense[(size_t)qg * kMf3DenseLd + (tile0 + tile) * kMfBM + rloc] = ok ? acc[mt][nt][reg] : -INFINITY;
          }
        } else {
          float m = acc[mt][nt][0];
#pragma unroll
          for (int reg = 1; reg < 16; ++reg) m = fmaxf(m, acc[mt][nt][reg]);
          if (__ballot(m >= thr[nt]) != 0ull) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const float v = acc[mt][nt][reg];
              bool pass = v >= thr[nt];
              if (pass) {
                const int row = (int)tile_row0 + rbase + (reg & 3) + 8 * (reg >> 2);
                if (needs_mask) pass = (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
                if (pass) {
                  const int pos = atomicAdd(&cand_cnt[qg], 1);
                  if (pos < cap) { cand_s[(size_t)qg * cap + pos] = v; cand_i[(size_t)qg * cap + pos] = row; }
                }
              }
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
      }
    }
    ++tile;
  }
}

}  // namespace vdbk
