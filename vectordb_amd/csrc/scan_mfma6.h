// scan_mfma6.h — the wide-tile pass kernel (scan_mfma5.h: 256 corpus rows x 256 queries per block, the same staging
// ring, barriers, counted waits, threshold/append epilogue) on v_mfma_f32_16x16x32_bf16 instead of 32x32x16.
//
// Why: the two shapes do the same FLOPs per cycle, but on REAL operands (random bit patterns) the chip holds a higher
// clock under the 16x16x32 stream: bare loops on random unit-vector-sized values, every CU busy, measured on MI355X
// (probes.h, vdb_get_stat probe_mfma_bf16_gflops / probe_mfma16_bf16_gflops): 32x32x16 1.80 PFLOP/s, 16x16x32 2.11
// (+17 %); on constant operands both run 2.4.  The pass kernel is bound by operand delivery in SHADER CLOCKS
// (scan_mfma5.h), so a higher clock speeds up the staging as well as the matrix pipe.
//
// What changes against scan_mfma5.h:
//   * a wave still owns 128 rows x 64 queries = 128 accumulator registers, now as 8 (M) x 4 (N) blocks of 16 x 16;
//   * a K-step (K = 64) is two k32 sub-steps = 16 POSITIONS (k32 sub-step t, row block mt) of 4 MFMAs each (the four
//     query blocks).  A fragments roll through a ring of four registers sets, read three positions (192 cycles)
//     ahead; the four B fragments of a k32 are double-buffered and read during the second half of the previous
//     k32: 48 fragment registers, as in scan_mfma5.h (whole-phase double buffering needs 64 and spills).  The last
//     four positions last 256 cycles, exactly the last k16 sub-step of scan_mfma5.h, so barrier, counted waits and
//     staging-load placement carry over one to one;
//   * fragment addressing: lane l reads row (l & 15) of its block, 16-byte chunk 4t + (l >> 4) of the stage's 128-byte
//     row; with the image's chunk swizzle (chunk ^ ((row >> 1) & 7)) the 16 lanes of a ds_read_b128 phase fall in 16
//     different bank groups (conflict-free, like the 32-row pattern);
//   * C layout: lane l holds query (l & 15) of a block and rows 4 (l >> 4) .. +3: one threshold per lane and N-block
//     (4), the max tree runs over the 32 scores a lane holds for one N-block, then per 4-score group.
//
// THE ISSUE ORDER OF A K-STEP IS WRITTEN OUT BY HAND (round 2, late).  Three things hipcc did to the first form of this
// kernel, found in its assembly, cost 10-13 % of a batch together (profiles/r02_kstep_schedule_ab.json):
//   * sched_group_barrier pins fix the PATTERN of a position (MFMA, read, MFMA, read, MFMA, load, MFMA) but not WHICH
//     read fills a slot: the scheduler sank every fragment read to just before the MFMA that needs it, so the three-
//     position look-ahead existed in the source only.  Now every MFMA slot ends in __builtin_amdgcn_sched_barrier(0) and
//     the instruction stream is the source order;
//   * the builtin LDS-DMA load (__builtin_amdgcn_global_load_lds) is a FLAT-encoded instruction with an LDS memory
//     operand; while one is in flight SIInsertWaitcnts ("pending flat") makes EVERY wait it inserts lgkmcnt(0), so a read
//     issued three positions ahead was waited for together with the one issued last.  The loads are inline asm here
//     (scalar base + one per-lane 32-bit offset, M0 set beside them): unknown to the compiler, they leave its counted
//     lgkmcnt(N) waits intact; vmcnt was counted by hand already;
//   * values only the epilogue needs (the four query numbers, the threshold patterns, 32 row numbers and their 32
//     `row < n_rows` compares, each an SGPR pair) were hoisted above the K loop or to the top of every epilogue and
//     spilled (12 VGPRs, 44 SGPRs through v_writelane, two reloads with s_waitcnt vmcnt(0) inside the loop, which drain
//     the staging ring).  They are derived where they are used, from values made opaque with empty asm: no spill is left.
#pragma once
#include "scan_mfma5.h"
#pragma clang diagnostic ignored "-Winline-asm"   // m0 on a clobber list (it is reserved; nothing else in these kernels uses it)

namespace vdbk {

typedef float f32x4m_t __attribute__((ext_vector_type(4)));

// NB = query blocks of 16 per wave: 4 = the 256-wide query tile described above; 2 = the same kernel on a 128-wide query
// tile (a wave owns 128 rows x 32 queries, a position is 2 MFMAs, a query stage is 16 KiB = two pieces per wave; same LDS
// layout with half of each query slot unused) for batches of 65...128 queries, where a 256-wide tile would be half padding.
template <int MODE, int ABL = 0, int NB = 4>   // ABL (diagnostic, results invalid): 1 = no staging loads, 2 = no fragment reads / MFMAs,
                                   // 3 = in-kernel cycle stamps (per wave: load wait, barrier wait, rest, epilogue) written to `dense`
__global__ __launch_bounds__(kM5Threads) void scan_mfma_wide16_kernel(
    const unsigned char* __restrict__ rows, int n_rows, const unsigned char* __restrict__ qb16, int b, int n_qtiles,
    int n_splits, int tiles_per_split, int n_pass_tiles, int tile_stride, int self0, int xcd_map,
    const float* __restrict__ thr_in, float* __restrict__ dense, int* __restrict__ cand_cnt,
    float* __restrict__ cand_s, int* __restrict__ cand_i, int cap, int ablate, int skip_n) {
  extern __shared__ __align__(16) unsigned char smem[];   // the only LDS object: 3 corpus + 2 query slots
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int l16 = lane & 15, c4 = lane >> 4;   // fragment row / K-chunk of a k32; C: query / row group

  int qt, sp;
  if (xcd_map) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    qt = j % n_qtiles;
    sp = xcd * (n_splits >> 3) + j / n_qtiles;
  } else {
    qt = blockIdx.x % n_qtiles;
    sp = blockIdx.x / n_qtiles;
  }
  constexpr int kBN = 64 * NB;   // queries per block
  // Query slots: two of 32 KiB for the 256-wide tile (all of LDS with the three corpus slots).  The 128-wide tile's
  // stages are 16 KiB, so THREE fit (144 KiB in all) and its query stages run one further ahead: its K-steps are half as
  // long (32 MFMAs per wave), and a query stage issued behind one barrier and needed at the next had ~0.6 us to land.
  constexpr int kSlotsB = NB == 2 ? 3 : 2;
  constexpr int kSlotB = NB == 2 ? kM5Slot / 2 : kM5Slot;
  const int qbase = qt * kBN;
  const int tile0 = sp * tiles_per_split;
  const int tile1 = min(tile0 + tiles_per_split, n_pass_tiles);
  const int my_tiles = tile1 > tile0 ? tile1 - tile0 : 0;
  if (my_tiles == 0) return;

  float thr[NB];
#pragma unroll
  for (int nt = 0; nt < NB; ++nt) thr[nt] = -INFINITY;
  if (MODE == 1 && thr_in) {
    const int q0 = qbase + wn * (16 * NB) + l16;
#pragma unroll
    for (int nt = 0; nt < NB; ++nt) thr[nt] = thr_in[min(q0 + 16 * nt, b - 1)];
  }
#pragma unroll
  for (int nt = 0; nt < NB; ++nt) asm volatile("" :: "v"(thr[nt]));   // retire the loads before the loop (see scan_mfma3.h)
  // The epilogue's filter runs on the scores' BIT PATTERNS as signed integers (scan_mfma5.h): against a positive
  // threshold "some score >= thr" is "max pattern >= thr's pattern"; a lane whose threshold is not positive gets
  // INT_MIN, so its filter always fires and the float compare decides.  Derived where it is used.

  // staging: LDS position L (16-byte units) of an operand image <- row L>>3, global chunk (L&7)^swz; corpus and
  // query tiles have the same shape, and piece i of a stage (L = i*512 + tid) lies 64 rows behind piece i-1 with
  // the same chunk, so ONE per-lane offset serves all eight loads of a stage (the piece goes into the scalar base).
  const unsigned g_off0 = (unsigned)((tid >> 3) * kMfRowBytes + (((tid & 7) ^ ((tid >> 4) & 7)) * 16));
  const int wbase16 = w * 1024;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto glds_asm = [&](const unsigned char* sbase, unsigned voff, unsigned lds_dst) {   // 64 lanes x 16 B -> LDS lds_dst + 16 lane
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
  };
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i >= i0 && i < i1) glds_asm(src + i * (64 * kMfRowBytes), g_off0, lds0 + slot * kM5Slot + wbase16 + i * (kM5Threads * 16));
  };
  auto stage_b = [&](int slot, int s_abs, int i0, int i1) {
    if (ABL == 1) return;
    const unsigned char* src = qblk + (s_abs >= kMfSteps ? s_abs - kMfSteps : s_abs) * (kMfBK * 2);
#pragma unroll
    for (int i = 0; i < NB; ++i)   // a query stage is NB pieces per wave
      if (i >= i0 && i < i1) glds_asm(src + i * (64 * kMfRowBytes), g_off0, lds0 + kM5BaseB + slot * kSlotB + wbase16 + i * (kM5Threads * 16));
  };

  // fragment reads: k32 sub-step t uses chunk 4t + c4 of rows l16 + 16*block
  const int swz = (l16 >> 1) & 7;
  const int a_row_off = (wm * 128 + l16) * 128;
  const int b_row_off = kM5BaseB + (wn * (16 * NB) + l16) * 128;
  int ck[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) ck[t] = ((4 * t + c4) ^ swz) << 4;

  // Fragments: a K-step is 16 POSITIONS p = 8 t + mt (k32 sub-step t, row block mt), 4 MFMAs each (the four query
  // blocks).  A fragments roll through a ring of four (position p uses fA[p & 3]; its read is issued three positions,
  // 192 cycles, ahead); the four B fragments of a k32 are double-buffered (fB[t & 1]), read one per position during
  // the second half of the previous k32.  48 fragment registers, as in scan_mfma5.h.
  bf16x8_t fA[4], fB[2][NB];
  auto read_a1 = [&](int abuf, int aslot, int t, int mt) {
    if (ABL == 2) return;
    fA[abuf] = *reinterpret_cast<const bf16x8_t*>(smem + aslot * kM5Slot + a_row_off + ck[t] + mt * 2048);
  };
  auto read_b1 = [&](int bbuf, int bslot, int t, int nt) {
    if (ABL == 2) return;
    fB[bbuf][nt] = *reinterpret_cast<const bf16x8_t*>(smem + bslot * kSlotB + b_row_off + ck[t] + nt * 2048);
  };

  f32x4m_t acc[8][NB];
  auto mfma1 = [&](int abuf, int bbuf, int mt, int nt, bool first) {
    if (ABL == 2) return;
    f32x4m_t c = acc[mt][nt];
    if (first) c = f32x4m_t{0.f, 0.f, 0.f, 0.f};
    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fA[abuf], fB[bbuf][nt], c, 0, 0, 0);
  };

  // prologue: stage 0 lands; stage 1 and the corpus half of stage 2 go out; sub-step 0 fragments are read
  stage_b(0, 0, 0, NB);
  stage_a(0, 0, 0, 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  stage_b(1, 1, 0, NB);
  stage_a(1, 1, 0, 4);
  if (kSlotsB == 3) stage_b(2, 2, 0, NB);
  read_a1(0, 0, 0, 0);
  read_a1(1, 0, 0, 1);
  read_a1(2, 0, 0, 2);
#pragma unroll
  for (int nt = 0; nt < NB; ++nt) read_b1(0, 0, 0, nt);

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
    // keep ks opaque: as a literal it is folded into per-step source addresses that are then all kept alive
    asm volatile("" : "+s"(ks));
    constexpr int AS = decltype(as_tag)::value;
    constexpr int BS = kSlotsB == 3 ? decltype(as_tag)::value : decltype(bs_tag)::value;   // stage ks sits in query slot ks % kSlotsB
    constexpr bool first = decltype(first_tag)::value != 0;
    // Staging loads are spread.  The texture addresser takes one LDS-DMA instruction per ~20 cycles per CU
    // and the issuing wave waits its turn: eight waves issuing eight each right behind the barrier stood
    // ~1000 cycles per step in that queue with no MFMA issued (in-kernel stamps, ABL = 3).  So: the corpus
    // stage ks+2 here, one load per position of sub-step 0's first half (its slot was released at the previous
    // barrier and it is needed only at the barrier after next); the query stage ks+2 behind this step's barrier.
    // Other placements (2+2 over two sub-steps, sub-step 1, 3/2/1/2) measured the same or worse.
    constexpr int A2 = (AS + 2) % kM5SlotsA;   // the corpus slot released at the previous step's barrier
    constexpr int AN = (AS + 1) % kM5SlotsA;   // next stage's slots (readable behind this step's barrier)
    // positions 0..11: this stage only.  Position p = MFMA, A fragment of position p + 3, MFMA, (positions 4..7: a B
    // fragment of k32 #1), MFMA, (positions 0..3: one load of the corpus stage ks+2), MFMA - in THIS order: every
    // slot is fenced (see the head of this file).
#pragma unroll
    for (int p = 0; p < 12; ++p) {
      const int pa = p + 3;
      const bool zero = first && p < 8;
#pragma unroll
      for (int nt = 0; nt < NB; ++nt) {
        mfma1(p & 3, p >> 3, p & 7, nt, zero);
        if (nt == 0) read_a1(pa & 3, AS, pa >> 3, pa & 7);
        if (nt == 1 && p >= 4 && p < 4 + NB) read_b1(1, BS, 1, p - 4);
        if (nt == (NB == 4 ? 2 : 1) && p < 4) stage_a(A2, ks + 2, p, p + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    unsigned long long t1 = 0, t2 = 0;
    if (ABL == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); t1 = __builtin_readcyclecounter(); }
    // All but my newest four loads (corpus stage ks+2, issued in this step) have landed.  No wait for fragment reads:
    // every read of the slot about to be refilled (the query slot) was issued before position 8 and has returned -
    // the MFMAs of position 11 waited for a read issued after them.
    // (The first step of a tile follows an epilogue whose stores and atomics share the counter and may complete out of
    // order with the loads: it drains to zero.  Loads return in order AMONG LOADS, so "at most 4 outstanding" with four
    // younger loads issued would prove the older ones landed anyway - but dropping the drain gained nothing, measured.)
    // (With three query slots the query stage issued behind the previous barrier may still be in flight as well.)
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (kSlotsB == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
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
    // positions 12..15 (behind the barrier): the next stage's first three A fragments and its k32 #0 B fragments; the
    // query stage ks+2 into the slot just released (one load each)
#pragma unroll
    for (int p = 12; p < 16; ++p) {
      const int pa = p + 3;
#pragma unroll
      for (int nt = 0; nt < NB; ++nt) {
        mfma1(p & 3, 1, p & 7, nt, false);
        if (nt == 0) { if (pa < 16) read_a1(pa & 3, AS, 1, pa & 7); else read_a1(pa & 3, AN, 0, pa - 16); }
        if (nt == 1 && p - 12 < NB) read_b1(0, (BS + 1) % kSlotsB, 0, p - 12);
        if (NB == 4 && nt == 2) stage_b(BS, ks + kSlotsB, p - 12, p - 11);
        if (NB == 2 && nt == 1 && p >= 14) stage_b(BS, ks + kSlotsB, p - 14, p - 13);   // the two positions without a B read
        __builtin_amdgcn_sched_barrier(0);
      }
    }
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
    unsigned long long te0 = 0;
    if (ABL == 3) te0 = __builtin_readcyclecounter();
    const long long tile_row0 = (long long)g_cur * kMfBM;
    ta = tn;
    g_cur = g_nxt;
    if (tile + 2 < my_tiles) { g_nxt = pass_tile_next(g_nxt, tile_stride, skip_n); tn = rows + (size_t)g_nxt * kTileBytes; }
    if (ablate & 4) continue;   // ablate bit 2: no epilogue (diagnostic, results invalid)

    // ---------------- epilogue of pass-local tile (tile0 + tile) ----------------
    const bool needs_mask = (tile_row0 + kMfBM > n_rows) || (qbase + kBN > b) || (self0 >= 0);
#pragma unroll
    for (int nt = 0; nt < NB; ++nt) {
      int l16e = l16;
      asm volatile("" : "+v"(l16e));   // derived per tile: hoisted, qg and what follows from it live across the K loop
      const int qg = qbase + wn * (16 * NB) + nt * 16 + l16e;
      if (MODE == 0) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int rloc = wm * 128 + mt * 16 + 4 * c4 + reg;
            const int row = (int)tile_row0 + rloc;
            const bool ok = (row < n_rows) && !(self0 >= 0 && row == self0 + qg);
            if (qg < b) dense[(size_t)qg * kMf3DenseLd + (tile0 + tile) * kMfBM + rloc] = ok ? acc[mt][nt][reg] : -INFINITY;
          }
      } else {
        int m8[8];   // bit patterns of the row blocks' maxima (4 scores each)
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
          m8[mt] = max(max(__float_as_int(acc[mt][nt][0]), __float_as_int(acc[mt][nt][1])),
                       max(__float_as_int(acc[mt][nt][2]), __float_as_int(acc[mt][nt][3])));
        const int m = max(max(max(m8[0], m8[1]), max(m8[2], m8[3])), max(max(m8[4], m8[5]), max(m8[6], m8[7])));
        float thr_e = thr[nt];
        asm volatile("" : "+v"(thr_e));   // likewise: the four patterns are not worth four registers across the K loop
        const int tb = thr_e > 0.f ? __float_as_int(thr_e) : (int)0x80000000;
        if (__ballot(m >= tb) != 0ull) {
#pragma unroll
          for (int mt = 0; mt < 8; ++mt) {
            if (__ballot(m8[mt] >= tb) == 0ull) continue;
            // The row numbers are derived HERE, behind the branch and from an opaque value: computable from loop
            // invariants, hipcc otherwise hoists all 32 of them and their 32 `row < n_rows` compares (each an SGPR
            // pair spilled through v_writelane) to the top of every epilogue, hit or no hit - 12 % of the kernel.
            int rbase = wm * 128 + mt * 16 + 4 * c4;
            asm volatile("" : "+v"(rbase));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[mt][nt][e];
              const int row = (int)tile_row0 + rbase + e;
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
    if (ABL == 3) prof_d += __builtin_readcyclecounter() - te0;   // the epilogue's share of "rest"
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (unused) staging loads
  if (MODE == 1) flush_pending();
  if (ABL == 3 && lane == 0) {
    float* o = dense + (size_t)(blockIdx.x * 8 + w) * 4;
    o[0] = (float)prof_a; o[1] = (float)prof_b; o[2] = (float)prof_c; o[3] = (float)prof_d;
  }
}

}  // namespace vdbk
