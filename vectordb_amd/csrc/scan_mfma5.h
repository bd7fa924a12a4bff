// scan_mfma5.h — wide-tile form of the batched pass kernel: 256 corpus rows x 256 queries
// per block.
//
// Why: the 256x128 kernels (scan_mfma3.h / scan_mfma4.h) stage (1/128 + 1/256) bytes per
// multiply-add column pair through LDS — 184 GB for 10M x 768 against 1024 queries — and the
// L2 -> LDS path delivers at most ~56 GB/s per CU (14.3 TB/s chip-wide; benchmarks/
// lds_dma_probe.hip): 12.9 ms of operand delivery, longer than the MFMA work itself.  A
// 256x256 tile stages 123 GB (8.6 ms) and halves the LDS fragment reads per MFMA (a wave owns
// 128 x 64: 6 ds_read_b128 feed 8 MFMAs instead of 4 feeding 4).
//
// The same probe sets the shape of the staging ring.  That rate needs loads in flight ACROSS the
// block barrier: a two-buffer kernel that refills right behind its barrier and drains before the
// next one (64 KiB burst, then nothing) gets 39.8 GB/s per CU; with 32 KiB still in flight at
// every barrier it is 56.  Half-line (64 B per row) pieces, i.e. K = 32 stages, get 37: the
// stage stays K = 64.  Three whole 64 KiB stages do not fit in 160 KiB of LDS, so the operands
// ring separately: THREE 32 KiB corpus slots and TWO 32 KiB query slots (160 KiB, all of it).
//
// Structure (512 threads = 8 waves, 2(M) x 4(N), two waves per SIMD, <= 256 registers):
//   * step t multiplies corpus slot t%3 by query slot t%2; the 12 K-steps of a tile are
//     unrolled (slots, the first step's C = 0 instead of a 128-register clear: compile-time);
//   * ONE block barrier per K-step, before the last of its four k16 sub-steps.  Each wave first
//     waits until only its newest 4 loads (corpus stage t+2, issued in this step) are
//     outstanding, so the barrier certifies that stage t+1 has landed and that everybody's reads
//     of stage t have returned; behind it the wave refills the query slot just released (stage
//     t+2); the corpus slot released there is refilled (stage t+3) in the next step's first
//     sub-step.  Issue order B(t+1), A(t+2), B(t+2), A(t+3) ... makes the counted wait exact;
//   * fragments are double-buffered in registers: the reads of sub-step j+1 (across the stage
//     boundary too) are issued between the MFMAs of sub-step j, so no MFMA waits on LDS;
//   * epilogue: VALU time is what it costs (two waves share a SIMD and reach it together), so it
//     is a max tree: per 32x32 block the lane's 16 scores reduce to four group maxima and one
//     block maximum; one ballot per block, then one per hot group of four, and only those four
//     scores are examined (the trees run on the scores' bit patterns, see thr_bits below).  A hit
//     waits in its lane's registers (two places) until some lane needs a third; then all waiting
//     hits reserve their slots at once (one atomic round trip per flush, not per hit).
//
// Where the time goes (10M x 768, B = 1024, final pass; rocprofv3 PMC, cycles per K-step and CU):
// MFMA alone 2100 (97 % of the matrix pipe, 2.0 GHz), staging alone 2580 (25 B/clk/CU through the
// texture path: the binding rate is in shader clocks), both 2920 at ~1.7 GHz.  The kernel is
// bound by operand delivery into LDS, 13 % above that floor.
// Thresholds, append buffers, exactness argument and fallback are those of scan_mfma3.h.
#pragma once
#include <type_traits>
#include "scan_mfma3.h"

namespace vdbk {

constexpr int kM5BN = 256;                          // queries per block
constexpr int kM5Slot = kMfBM * kMfBK * 2;          // 32 KiB: 256 rows x 128 B, either operand
constexpr int kM5SlotsA = 3;
constexpr int kM5SlotsB = 2;
constexpr int kM5BaseB = kM5SlotsA * kM5Slot;
constexpr int kM5Threads = 512;
static_assert(kM5BN == kMfBM, "one set of staging offsets serves both operands");

__host__ __device__ constexpr size_t m5_smem_bytes() { return (size_t)(kM5SlotsA + kM5SlotsB) * kM5Slot; }   // 160 KiB

template <int MODE, int ABL = 0>   // ABL (diagnostic, results invalid): 1 = no staging loads, 2 = no fragment reads / MFMAs,
                                   // 3 = in-kernel cycle stamps (per wave: load wait, barrier wait, rest) written to `dense`
__global__ __launch_bounds__(kM5Threads) void scan_mfma_wide_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int ablate, int skip_n) {
  extern __shared__ __align__(16) unsigned char smem[];   // the only LDS object: 3 corpus + 2 query slots
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
  // The epilogue's filter runs on the scores' BIT PATTERNS as signed integers: against a positive threshold
  // (the rule: it is the k'-th best cosine of a sample) "some score >= thr" is "max pattern >= thr's pattern" —
  // a positive float orders like its pattern, every negative float is a negative integer.  Integer max has no
  // NaN rule: 10 ops per 16 scores where fmaxf costs 19 (a canonicalising v_max x,x in front of every input).
  // A lane whose threshold is not positive gets INT_MIN: its filter always fires and the float compare decides.
  int thr_bits[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) thr_bits[nt] = thr[nt] > 0.f ? __float_as_int(thr[nt]) : (int)0x80000000;

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
  constexpr size_t kTileBytes = (size_t)kMfBM * kMfRowBytes;
  int g_cur = pass_tile(tile0, tile_stride, skip_n);             // corpus tile being multiplied (scan_mfma.h)
  int g_nxt = my_tiles > 1 ? pass_tile_next(g_cur, tile_stride, skip_n) : g_cur;   // the one after it (the last tile names
  const unsigned char* ta = rows + (size_t)g_cur * kTileBytes;                     // itself: its trailing stage loads land
  const unsigned char* tn = rows + (size_t)g_nxt * kTileBytes;                     // in slots nobody reads)
  // This wave's four 1 KiB pieces of a stage.  s_abs = K-step index counted from the current
  // tile's first (12..14 = the next tile's first three).
  auto stage_a = [&](int slot, int s_abs, int i0, int i1) {
    if (ABL == 1) return;
    const bool wrap = s_abs >= kMfSteps;
    const unsigned char* src = (wrap ? tn : ta) + (wrap ? s_abs - kMfSteps : s_abs) * (kMfBK * 2);
    unsigned char* st = smem + slot * kM5Slot + wbase16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= i0 && i < i1) mf_glds16(src + g_off[i], st + i * (kM5Threads * 16));
  };
  auto stage_b = [&](int slot, int s_abs, int i0, int i1) {
    if (ABL == 1) return;
    const unsigned char* src = qblk + (s_abs >= kMfSteps ? s_abs - kMfSteps : s_abs) * (kMfBK * 2);
    unsigned char* st = smem + kM5BaseB + slot * kM5Slot + wbase16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= i0 && i < i1) mf_glds16(src + g_off[i], st + i * (kM5Threads * 16));
  };

  // fragment reads: k16 sub-step s uses chunk 2s+hsel of rows (lane&31) + 32*block
  const int swz = ((lane & 31) >> 1) & 7;
  const int a_row_off = (wm * 128 + (lane & 31)) * 128;
  const int b_row_off = kM5BaseB + (wn * 64 + (lane & 31)) * 128;
  int ck[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) ck[s] = ((2 * s + hsel) ^ swz) << 4;

  bf16x8_t fA[2][4], fB[2][2];
  auto read_frags = [&](int fbuf, int aslot, int bslot, int s) {
    if (ABL == 2) return;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      fA[fbuf][mt] = *reinterpret_cast<const bf16x8_t*>(smem + aslot * kM5Slot + a_row_off + ck[s] + mt * 4096);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      fB[fbuf][nt] = *reinterpret_cast<const bf16x8_t*>(smem + bslot * kM5Slot + b_row_off + ck[s] + nt * 4096);
  };

  f32x16_t acc[4][2];
  auto mfma8 = [&](int fbuf, bool first) {
    if (ABL == 2) return;
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
  // one k16 sub-step = 8 MFMAs; between them go the next sub-step's 6 fragment reads and, behind the
  // barrier, the step's 8 staging loads
  auto pin_schedule = [&](int nv) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            // 1 MFMA
      if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                 // 1 DS read
      if ((nv == 4 && (i & 1)) || (nv == 3 && (i == 1 || i == 4 || i == 7)) || (nv == 2 && (i == 3 || i == 7)) || (nv == 1 && i == 5))
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);                          // 1 VMEM (LDS-DMA)
    }
  };

  // prologue: stage 0 lands; stage 1 and the corpus half of stage 2 go out; sub-step 0 fragments are read
  stage_b(0, 0, 0, 4);
  stage_a(0, 0, 0, 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stage_b(1, 1, 0, 4);
  stage_a(1, 1, 0, 4);
  read_frags(0, 0, 0, 0);

  // A hit waits in its lane's registers (kPend places per lane) until some lane needs one more; then every
  // waiting hit reserves its slot in its query's candidate buffer at once.  The returning atomic (~1 us, and
  // its s_waitcnt drains the staged loads in flight too) is thus paid once per flush, not once per hit: with
  // one place the 1 %-hit-rate sample pass flushed at almost every hot register and ran atomics-bound.
  constexpr int kPend = 2;
  float pend_v[kPend];
  int pend_row[kPend], pend_q[kPend];
  int pend_n = 0;
#pragma unroll
  for (int s = 0; s < kPend; ++s) { pend_v[s] = 0.f; pend_row[s] = 0; pend_q[s] = 0; }
  auto flush_pending = [&]() {
    int pos[kPend];
#pragma unroll
    for (int s = 0; s < kPend; ++s) pos[s] = s < pend_n ? atomicAdd(&cand_cnt[pend_q[s] * kMfCntStride], 1) : cap;   // all in flight, one wait
#pragma unroll
    for (int s = 0; s < kPend; ++s)
      if (pos[s] < cap) { cand_s[(size_t)pend_q[s] * cap + pos[s]] = pend_v[s]; cand_i[(size_t)pend_q[s] * cap + pos[s]] = pend_row[s]; }
    pend_n = 0;
  };
  unsigned long long prof_a = 0, prof_b = 0, prof_c = 0, prof_d = 0, prof_last = 0;
  // one K-step on corpus slot AS and query slot BS (compile-time); ks is a scalar
  auto step = [&](auto as_tag, auto bs_tag, auto first_tag, int ks) {
    // keep ks opaque: as a literal it is folded into forty-odd per-lane 64-bit source offsets that spill
    asm volatile("" : "+s"(ks));
    constexpr int AS = decltype(as_tag)::value;
    constexpr int BS = decltype(bs_tag)::value;
    constexpr bool first = decltype(first_tag)::value != 0;
    // Staging loads are spread.  The texture addresser takes one LDS-DMA instruction per ~20 cycles per CU
    // and the issuing wave waits its turn: eight waves issuing eight each right behind the barrier stood
    // ~1000 cycles per step in that queue with no MFMA issued (in-kernel stamps, ABL = 3).  So: the corpus
    // stage ks+2 here, one load per two MFMAs of sub-step 0 (its slot was released at the previous barrier
    // and it is needed only at the barrier after next); the query stage ks+2 behind this step's barrier.
    // Other placements (2+2 over two sub-steps, sub-step 1, 3/2/1/2) measured the same or worse.
    constexpr int A2 = (AS + 2) % kM5SlotsA;   // the corpus slot released at the previous step's barrier
    read_frags(1, AS, BS, 1);
    stage_a(A2, ks + 2, 0, 4);
    mfma8(0, first);
    pin_schedule(4);
    read_frags(0, AS, BS, 2);
    mfma8(1, false);
    pin_schedule(0);
    read_frags(1, AS, BS, 3);
    mfma8(0, false);
    pin_schedule(0);
    unsigned long long t1 = 0, t2 = 0;
    if (ABL == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t1 = __builtin_readcyclecounter(); }
    // all but my newest four loads (corpus stage ks+2, issued in this step) have landed, every read of
    // this stage has returned.
    // (The first step of a tile follows an epilogue whose stores and atomics share the counter and may
    // complete out of order with the loads: there the count proves nothing, so it drains to zero.)
    if (first) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    if (ABL == 3) t2 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ABL == 3) {
      const unsigned long long t3 = __builtin_readcyclecounter();
      if (prof_last) prof_c += t1 - prof_last;
      prof_a += t2 - t1;
      prof_b += t3 - t2;
      prof_last = t3;
    }
    stage_b(BS, ks + 2, 0, 4);                // the query slot just released
    if (ABL == 3) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t4 = __builtin_readcyclecounter();
      asm volatile("" :: "s"(t4));
      prof_d += t4 - prof_last;
      __builtin_amdgcn_sched_barrier(0);
    }
    read_frags(0, (AS + 1) % kM5SlotsA, BS ^ 1, 0);   // first fragments of the next stage
    mfma8(1, false);
    pin_schedule(4);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  for (int tile = 0; tile < my_tiles; ++tile) {
    step(I0{}, I0{}, I1{}, 0);
    step(I1{}, I1{}, I0{}, 1);
    step(I2{}, I0{}, I0{}, 2);
    step(I0{}, I1{}, I0{}, 3);
    step(I1{}, I0{}, I0{}, 4);
    step(I2{}, I1{}, I0{}, 5);
    step(I0{}, I0{}, I0{}, 6);
    step(I1{}, I1{}, I0{}, 7);
    step(I2{}, I0{}, I0{}, 8);
    step(I0{}, I1{}, I0{}, 9);
    step(I1{}, I0{}, I0{}, 10);
    step(I2{}, I1{}, I0{}, 11);
    const long long tile_row0 = (long long)g_cur * kMfBM;
    ta = tn;
    g_cur = g_nxt;
    if (tile + 2 < my_tiles) { g_nxt = pass_tile_next(g_nxt, tile_stride, skip_n); tn = rows + (size_t)g_nxt * kTileBytes; }
    if (ablate & 4) continue;   // ablate bit 2: no epilogue (diagnostic, results invalid)

    // ---------------- epilogue of pass-local tile (tile0 + tile) ----------------
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
          int m4[4];   // bit patterns of the four group maxima
#pragma unroll
          for (int g = 0; g < 4; ++g)
            m4[g] = max(max(__float_as_int(acc[mt][nt][4 * g]), __float_as_int(acc[mt][nt][4 * g + 1])),
                        max(__float_as_int(acc[mt][nt][4 * g + 2]), __float_as_int(acc[mt][nt][4 * g + 3])));
          const int m = max(max(m4[0], m4[1]), max(m4[2], m4[3]));
          if (__ballot(m >= thr_bits[nt]) != 0ull) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (__ballot(m4[g] >= thr_bits[nt]) == 0ull) continue;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float v = acc[mt][nt][4 * g + e];
                const int row = (int)tile_row0 + rbase + e + 8 * g;   // register 4g+e holds tile row rbase + e + 8g
                bool pass = v >= thr[nt];
                if (needs_mask) pass = pass && (row < n_rows) && (qg < b) && !(self0 >= 0 && row == self0 + qg);
                if (__ballot(pass) == 0ull) continue;
                if (__ballot(pass && pend_n == kPend) != 0ull) flush_pending();   // a lane has no place left
                if (pass) {
#pragma unroll
                  for (int s = 0; s < kPend; ++s)
                    if (pend_n == s) { pend_v[s] = v; pend_row[s] = row; pend_q[s] = qg; }
                  ++pend_n;
                }
              }
            }
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) staging loads
  if (MODE == 1) flush_pending();
  if (ABL == 3 && lane == 0) {
    float* o = dense + (size_t)(blockIdx.x * 8 + w) * 4;
    o[0] = (float)prof_a; o[1] = (float)prof_b; o[2] = (float)prof_c; o[3] = (float)prof_d;
  }
}

}  // namespace vdbk
