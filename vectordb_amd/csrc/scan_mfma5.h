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
//     every wave stages 8 KiB per K-step with 8 global_load_lds_dwordx4; the 12 K-steps of a
//     tile are unrolled, so stage parity, K offsets and the first step's zero accumulator
//     (MFMA with C = 0: no 128-register clear per tile) are compile-time;
//   * ONE block barrier per K-step, placed before the last of its four k16 sub-steps:
//     it certifies both that the next stage has landed (each wave waits for its own
//     loads first) and that everybody's reads of the current stage have returned, so the
//     current buffer is refilled for step+2 right behind it — a full step of flight time;
//   * fragments are double-buffered in registers: the reads of sub-step j+1 (across the
//     stage boundary too) are issued before the MFMAs of sub-step j, so no MFMA waits on LDS;
//   * epilogue: VALU time is what it costs (two waves share a SIMD and reach it together), so
//     it is a max tree: per 32x32 block the lane's 16 scores reduce to four group maxima and
//     one block maximum; one ballot per block, then one per hot group of four, and only those
//     four scores are examined.  Hits go to a per-wave LDS queue (three ds_write_b32) and are
//     moved to the candidate buffers 64 at a time, so the returning global atomic that
//     reserves a slot is paid per drain, not per hit.
// Thresholds, append buffers, exactness argument and fallback are those of scan_mfma3.h.
#pragma once
#include <type_traits>
#include "scan_mfma3.h"

namespace vdbk {

constexpr int kM5BN = 256;                          // queries per block
constexpr int kM5StageA = kMfBM * kMfBK * 2;        // 32 KiB
constexpr int kM5StageB = kM5BN * kMfBK * 2;        // 32 KiB
constexpr int kM5Stage = kM5StageA + kM5StageB;     // 64 KiB
constexpr int kM5NumStages = 2;
constexpr int kM5Threads = 512;

__host__ __device__ constexpr size_t m5_smem_bytes() { return (size_t)kM5NumStages * kM5Stage + 8 * 256 * 12; }   // + per-wave hit queues

template <int MODE, int LP>
__global__ __launch_bounds__(kM5Threads) void scan_mfma_wide_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int ablate) {
  extern __shared__ __align__(16) unsigned char smem[];   // the only LDS object: 2 stages + 8 hit queues
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
  if (my_tiles == 0) return;

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
  const size_t tile_bytes = (size_t)tile_stride * (kMfBM * kMfRowBytes);
  const unsigned char* ta = rows + (size_t)tile0 * tile_bytes;   // corpus tile being multiplied
  const unsigned char* tn = my_tiles > 1 ? ta + tile_bytes : ta; // the one after it (the last tile names itself:
                                                                 // its two trailing stage loads land in buffers nobody reads)
  // piece i (0..3 corpus, 4..7 queries; 1 KiB per wave) of stage s_abs = K-step index counted from
  // the current tile's first (12, 13 = the next tile's first two); 12 is even, so parity = buffer
  // `partner`: the same piece of the wave that shares this wave's SIMD (w ^ 4: 32 rows further on or back)
  const long long pd_g = wm == 0 ? 32 * kMfRowBytes : -32 * kMfRowBytes;
  const int pd_l = wm == 0 ? 4096 : -4096;
  auto piece = [&](int i, int s_abs, bool partner) {
    const bool wrap = s_abs >= kMfSteps;
    const long long koff = (wrap ? s_abs - kMfSteps : s_abs) * (kMfBK * 2) + (partner ? pd_g : 0);
    unsigned char* st = smem + (s_abs & 1) * kM5Stage + wbase16 + (partner ? pd_l : 0);
    if (i < 4) mf_glds16((wrap ? tn : ta) + koff + g_off[i], st + i * (kM5Threads * 16));
    else mf_glds16(qblk + koff + g_off[i - 4], st + kM5StageA + (i - 4) * (kM5Threads * 16));
  };

  // fragment reads: k16 sub-step s uses chunk 2s+hsel of rows (lane&31) + 32*block
  const int swz = ((lane & 31) >> 1) & 7;
  const int a_row_off = (wm * 128 + (lane & 31)) * 128;
  const int b_row_off = kM5StageA + (wn * 64 + (lane & 31)) * 128;
  int ck[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ck[s] = ((2 * s + hsel) ^ swz) << 4;

  bf16x8_t fA[2][4], fB[2][2];
  auto read_frags = [&](int fbuf, int stage_buf, int s) {
    const unsigned char* sb = smem + stage_buf * kM5Stage;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) fA[fbuf][mt] = *reinterpret_cast<const bf16x8_t*>(sb + a_row_off + ck[s] + mt * 4096);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) fB[fbuf][nt] = *reinterpret_cast<const bf16x8_t*>(sb + b_row_off + ck[s] + nt * 4096);
  };

  f32x16_t acc[4][2];
  auto mfma8 = [&](int fbuf, bool first) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x16_t c = acc[mt][nt];
        if (first) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
        }
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fA[fbuf][mt], fB[fbuf][nt], c, 0, 0, 0);
      }
  };
  // one k16 sub-step = 8 MFMAs; between them go the next sub-step's 6 fragment reads and NV staging loads
  auto pin_schedule = [&](auto nv_tag) {
    constexpr int NV = decltype(nv_tag)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // 1 MFMA
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // 1 DS read
      const int nv = ((i + 1) * NV) / 8 - (i * NV) / 8;                        // VMEM (LDS-DMA) after this MFMA
      if (nv == 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      if (nv == 2) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
    }
  };
  using NV0 = std::integral_constant<int, 0>;

  // per-wave hit queue in LDS, behind the two stages
  constexpr int kQCap = 256;
  float* q_s = reinterpret_cast<float*>(smem + kM5NumStages * kM5Stage + w * (kQCap * 12));
  int* q_r = reinterpret_cast<int*>(q_s + kQCap);
  int* q_q = q_r + kQCap;
  int qn = 0;   // wave-uniform
  auto drain = [&]() {
    for (int base = 0; base < qn; base += 64) {
      const int i = base + lane;
      if (i < qn) {
        const float s = q_s[i];
        const int r = q_r[i], q = q_q[i];
        const int pos = atomicAdd(&cand_cnt[q], 1);
        if (pos < cap) { cand_s[(size_t)q * cap + pos] = s; cand_i[(size_t)q * cap + pos] = r; }
      }
    }
    qn = 0;
  };

  // prologue: stage 0 lands, stage 1 starts, sub-step 0 fragments of stage 0 are read
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(i, 0, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(i, 1, false);
  read_frags(0, 0, 0);

  // one K-step; the stage parity `buf` and `first` (first step of a tile: C = 0) are compile-time,
  // the K-step index ks is a scalar
  auto step = [&](auto buf_tag, auto first_tag, int ks) {
    constexpr int buf = decltype(buf_tag)::value;
    constexpr bool first = decltype(first_tag)::value != 0;
    read_frags(1, buf, 1);
    mfma8(0, first);
    pin_schedule(NV0{});
    read_frags(0, buf, 2);
    mfma8(1, false);
    pin_schedule(NV0{});
    read_frags(1, buf, 3);
    mfma8(0, false);
    pin_schedule(NV0{});
    // every read of this stage has returned and my share of the next stage has landed -> barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // Refill the buffer just released, for step+2, and read the first fragments of the next stage.
    // An LDS-DMA instruction holds its wave's issue slot ~100 cycles while the L2->LDS path is busy,
    // and a wave stuck there issues no MFMAs.  With LP > 8 the two waves of a SIMD (w, w^4) take
    // turns: this step's loader issues LP of the pair's 16 pieces, its partner the other 16-LP, so
    // one of them keeps the matrix pipe fed while the other feeds the memory pipe.
    if (LP == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) piece(j, ks + 2, false);
      read_frags(0, buf ^ 1, 0);
      mfma8(1, false);
      pin_schedule(std::integral_constant<int, 8>{});
    } else {
      // (the branch holds only the loads: MFMAs inside it would make hipcc copy the accumulators at the join)
      if (wm == buf) {
#pragma unroll
        for (int j = 0; j < LP; ++j) piece(j & 7, ks + 2, j >= 8);
      } else {
#pragma unroll
        for (int j = LP - 8; j < 8; ++j) piece(j, ks + 2, false);
      }
      read_frags(0, buf ^ 1, 0);
      mfma8(1, false);
      pin_schedule(NV0{});
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  for (int tile = 0; tile < my_tiles; ++tile) {
    step(I0{}, I1{}, 0);
    step(I1{}, I0{}, 1);
    for (int ks = 2; ks < kMfSteps; ks += 2) {
      step(I0{}, I0{}, ks);
      step(I1{}, I0{}, ks + 1);
    }
    ta = tn;
    if (tile + 2 < my_tiles) tn += tile_bytes;
    if (ablate & 4) continue;   // ablate bit 2: no epilogue (diagnostic, results invalid)

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
          float m4[4];
#pragma unroll
          for (int g = 0; g < 4; ++g)
            m4[g] = fmaxf(fmaxf(acc[mt][nt][4 * g], acc[mt][nt][4 * g + 1]), fmaxf(acc[mt][nt][4 * g + 2], acc[mt][nt][4 * g + 3]));
          const float m = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
          if (__ballot(m >= thr[nt]) != 0ull) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (__ballot(m4[g] >= thr[nt]) == 0ull) continue;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = acc[mt][nt][4 * g + e];
                const int row = (int)tile_row0 + rbase + e + 8 * g;   // register 4g+e holds tile row rbase + e + 8g
                bool pass = v >= thr[nt];
                if (needs_mask) pass = pass && (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
                const unsigned long long pb = __ballot(pass);
                if (pb != 0ull) {
                  if (qn > kQCap - 64) drain();
                  const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pb, 0u));
                  if (pass) { q_s[slot] = v; q_r[slot] = row; q_q[slot] = qg; }
                  qn += __popcll(pb);
                }
              }
            }
          }
        }
      }
    }
    if (MODE == 1 && qn >= kQCap / 2) drain();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) staging loads
  if (MODE == 1) drain();
}

}  // namespace vdbk
