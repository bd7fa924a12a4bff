// scan_fused.h — the whole small-batch search (<= 4 queries, k' <= 64) in ONE launch: collection.query as the reference
// issues it (one query -> one result, src/vectordb.py:259-264) costs one kernel instead of four.
//   prologue : every wave L2-normalises the raw queries itself (768 floats each; the arithmetic of
//              normalize_rows_kernel, bit for bit) — no separate normalisation launch;
//   body     : scan_corpus_tiles (scan_gemv.h), unchanged;
//   epilogue : each block publishes its partial list, then draws a ticket; the block that draws the LAST one folds
//              the gridDim.x lists, re-scores the k' survivors in fp32 (rerank) and writes idx / dist — no fold and
//              no finish launch.  At an 8-GPU shard's 1.25M rows the three launches it replaces and their gaps were
//              ~30 us of a 0.32 ms step.
// Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md, in-launch split-K recipe):
// list stores (write-through, agent scope, for one query; plain + an agent release fence by lane 0 for two or four)
// -> every wave s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: vmcnt(0), relaxed agent fetch_add on the ticket; the
// last arriver: agent acquire fence, vmcnt(0), __syncthreads, plain loads.
// The ticket word (and the tuning words on the 128-byte lines behind it: scan-time sums and class shares, below) must
// be zero before the FIRST launch: zeroed once where they are allocated; the last arriver resets ticket and sums.
//
// The fold does not merge 512 lists one after the other (128 bitonic merges per wave ~ 60 us): it first takes a
// BOUND from the list heads — each wave the ceil(k'/4)-th largest head of its quarter of the lists, by bisection
// over the key bits with ballots; the minimum of the four is a score that at least k' candidates reach — then all
// 256 threads sweep the candidates once (16-byte loads) and keep those that reach the bound (about 1.5 k' on random
// data), which are sorted by bitonic wave sorts.  More survivors than the LDS buffer holds (adversarial orders,
// thousands of equal scores): the exact list-by-list fold instead.  Same candidate order as everywhere: score
// descending, row index ascending; the result is bit-identical to the unfused kernels'.
#pragma once
#include "launch.h"
#include "select_dev.h"

namespace vdbk {

constexpr int kFusedSurvivorCap = 2048;     // (score, index) pairs the last block keeps in LDS
static_assert(kFusedMaxParts == 4 * kScanThreads, "the bound step holds 4 list heads per thread");
constexpr int kFusedRerankRows = 10;        // rows a wave re-scores at a time (30 x 16 bytes in flight per lane at 768-d: k' = 40 in ONE round trip)
constexpr int kFusedTilesPerWaveAndRound = kFusedTilesPerWaveAndRoundC;   // super-round = this many tiles per wave of the grid (512 blocks: 16384 tiles = 200 MB of 768-d bf16)
constexpr int kFusedSweepAhead = 20;        // 16-byte loads per thread of one sweep round (512 lists x 40 = one round)

// LDS of the last block's epilogue, behind the block merge area (see fused_smem_bytes)
template <int DIM>
struct FusedTail {
  double inv[4];                             // 1 / |q| of the block's queries (the rerank's normalised query)
  int flag; int count; unsigned bound[4]; int pad[2];
  float ls[64]; int li[64];                  // the query's k' list
  float ms[2 * 64]; int mi[2 * 64];          // block_tree_merge scratch
  float q[DIM];                              // the normalised query (rerank)
  float sur_s[kFusedSurvivorCap]; int sur_i[kFusedSurvivorCap];
};

__host__ __device__ constexpr size_t fused_merge_bytes(int qb, int kk) { return ((size_t)kScanWaves * qb * kk * 8 + 15) / 16 * 16; }
template <int DIM>
__host__ __device__ constexpr size_t fused_smem_bytes(int qb, int kk) { return fused_merge_bytes(qb, kk) + sizeof(FusedTail<DIM>); }

// order-preserving map of a float's bits to unsigned (larger float <=> larger key)
__device__ __forceinline__ unsigned score_key(float s) {
  const unsigned u = __float_as_uint(s);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float key_score(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// fp32 re-score of the k' candidates in li[] against the query q (LDS) by the four waves of the last block: wave w takes
// candidates w, w + 4, ..., kFusedRerankRows of them in flight; per row the summation order of final_sorted_kernel
// (j = lane, lane + 64, ...; x, y, z, w), so both give the same bits.  Every row load is unconditional (an absent
// candidate reads row 0 and is dropped) and the fp32 / bf16 source is a template value: a branch around a load makes
// the compiler wait for every row on its own (11 us for 40 rows instead of one round trip).
template <int DIM, bool F32>
__device__ __forceinline__ void fused_rerank(float* ls, const int* li, const float* q, int kk,
                                             const float* __restrict__ rows_f32, const uint16_t* __restrict__ rows_b16) {
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  const f32x4* qv = reinterpret_cast<const f32x4*>(q);
  constexpr int C4 = DIM / 4;
  for (int c0 = w; c0 < kk; c0 += kScanWaves * kFusedRerankRows) {
    int ids[kFusedRerankRows], rid[kFusedRerankRows];
    float acc[kFusedRerankRows];
#pragma unroll
    for (int u = 0; u < kFusedRerankRows; ++u) {
      const int c = c0 + u * kScanWaves;
      ids[u] = c < kk ? li[c] : kIdxSentinel;
      rid[u] = ids[u] != kIdxSentinel ? ids[u] : 0;
      acc[u] = 0.f;
    }
#pragma unroll
    for (int j0 = 0; j0 < C4; j0 += 64) {
      const int j = min(j0 + lane, C4 - 1);
      const bool live = j0 + lane < C4;
      const f32x4 b = qv[j];
#pragma unroll
      for (int u = 0; u < kFusedRerankRows; ++u) {
        f32x4 r;
        if constexpr (F32) {
          r = reinterpret_cast<const f32x4*>(rows_f32 + (size_t)rid[u] * DIM)[j];
        } else {
          const uint2 p = reinterpret_cast<const uint2*>(rows_b16 + (size_t)rid[u] * DIM)[j];
          r.x = __uint_as_float(p.x << 16); r.y = __uint_as_float(p.x & 0xffff0000u);
          r.z = __uint_as_float(p.y << 16); r.w = __uint_as_float(p.y & 0xffff0000u);
        }
        if (live) {
          acc[u] = fmaf(r.x, b.x, acc[u]); acc[u] = fmaf(r.y, b.y, acc[u]);
          acc[u] = fmaf(r.z, b.z, acc[u]); acc[u] = fmaf(r.w, b.w, acc[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kFusedRerankRows; ++u) {
      const float tot = wave_sum(acc[u]);
      if (lane == 0 && ids[u] != kIdxSentinel) ls[c0 + u * kScanWaves] = tot;
    }
  }
}

template <typename T, int DIM, int QB, bool NT>
__global__ __launch_bounds__(kScanThreads) void scan_topk_fused_kernel(const FusedScanArgs a) {
  using GM = ScanGeom<T, DIM>;
  constexpr int EPC = GM::EPC, CPR = GM::CPR, L = GM::L;
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  const int kk = a.kk;
  auto stamp = [&](int i) { if (a.stamps && threadIdx.x == 0) a.stamps[i] = (long long)wall_clock64(); };
  stamp(16 + 6 * (int)blockIdx.x);
  if (a.stamps && threadIdx.x == 0) {   // where this block runs: XCC_ID (hwreg 20, 4 bits) and HW_ID (hwreg 4)
    a.stamps[20 + 6 * (int)blockIdx.x] = (long long)__builtin_amdgcn_s_getreg((3 << 11) | 20);
    a.stamps[21 + 6 * (int)blockIdx.x] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }

  // ---- prologue: 1 / (|q| + 1e-30) per query, as normalize_rows_kernel computes it (fp64 sum over c = lane, lane + 64, ...)
  double inv[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    inv[q] = 1.0;
    if (a.normalize) {   // uniform
      const float* x = a.q + q * DIM;
      double ss = 0.0;
      for (int c = lane; c < DIM; c += 64) { const double v = (double)x[c]; ss = fma(v, v, ss); }
      ss = wave_sum_f64(ss);
      inv[q] = 1.0 / (sqrt(ss) + 1e-30);
    }
  }
  float qreg[QB][L][EPC];
#pragma unroll
  for (int q = 0; q < QB; ++q)
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int cc = (j * 64 + lane) % CPR;
#pragma unroll
      for (int e = 0; e < EPC; ++e) qreg[q][j][e] = (float)((double)a.q[q * DIM + cc * EPC + e] * inv[q]);   // x * 1.0 is x
    }

  WaveTopK<1> tk[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) { tk[q].init(kk); if (a.probe) tk[q].thr_s = INFINITY; }   // probe: no row ever beats the threshold

  // ---- this wave's tiles.  The blocks of a launch do not stream at one rate (in-kernel stamps, profiles/r03_scan_stamps_*):
  // blocks b, b + 8, ... share an XCD, and with equal shares the odd XCDs of an MI355X need ~7 % longer than the even
  // ones (2170 vs 2030 us per 10M-row pass); and the SECOND block a CU receives (b >= the CU count) runs 2.5-4.5 %
  // behind the first.  The pass ends with the slowest block.  So the blocks form groups (b & 7, b / n_cu), each group
  // gets tiles in proportion to share[group] x its blocks — a slice of every super-round, so that the grid still walks
  // the corpus as one window (TileWalk, scan_gemv.h) — and the shares TUNE THEMSELVES: every
  // block adds its scan time to its group's sum, and the last block turns the sums into the next launch's shares
  // (below).  The partition is a function of the share table alone, so it is exact whatever the table holds.
  // (Tried first and measured slower than the uneven static split: claiming the last fifth of the tiles from work pools
  // with atomics, 0.2-7 % — each claim's round trip sits in the wave's in-order load queue; pooling a block's tiles over
  // its four waves through an LDS counter: no gain, the waves of a CU rise and fall together.)
  constexpr int RPT = GM::RPT;
  const long long n_tiles = ((long long)a.n_rows + RPT - 1) / RPT;
  const int n_blocks = (int)gridDim.x;
  const int n_cu = max(8, a.n_cu & ~7);
  const int n_groups = 8 * min(kFusedMaxRounds, (n_blocks + n_cu - 1) / n_cu);
  auto group_blocks = [&](int g) -> int {   // blocks b with (b & 7) == (g & 7) in [r n_cu, min(n_blocks, (r + 1) n_cu)), r = g >> 3 (the last round takes the rest)
    const int r = g >> 3, lo = r * n_cu, hi = (g >> 3) == (n_groups >> 3) - 1 ? n_blocks : min(n_blocks, lo + n_cu);
    return max(0, (hi - lo - (g & 7) + 7) >> 3);
  };
  const int my_round = min((int)blockIdx.x / n_cu, (n_groups >> 3) - 1);
  const int grp = ((int)blockIdx.x & 7) + 8 * my_round;
  TileWalk wk = TileWalk::plain(n_tiles, (long long)blockIdx.x * kScanWaves + w, (long long)n_blocks * kScanWaves);
  unsigned* tune = a.ticket + kFusedDurOffset;          // [32] scan-time sums, then (kFusedShareOffset) [32] shares
  const bool adapt = a.balance == 1 && !a.allow && !a.probe && n_tiles >= 32 * (long long)wk.nwv;   // long enough a pass that the times mean something
  if (a.balance) {
    long long before = 0, mine = 0, total = 0;
    for (int g = 0; g < n_groups; ++g) {
      const unsigned sh = a.balance == 1 ? a.ticket[kFusedShareOffset + g] : 0u;   // (2 = diagnostic: grouped ranges, equal shares)
      const long long wgt = (long long)(sh ? sh : kFusedShareUnit) * group_blocks(g);
      if (g < grp) before += wgt;
      if (g == grp) mine = wgt;
      total += wgt;
    }
    // super-rounds of a.super_rounds (default kFusedTilesPerWaveAndRound) tiles per wave of the grid: the class owns [S_g, S_g + q) of each;
    // what is left behind the last full one (all of a small store) is split by the same weights (TileWalk, scan_gemv.h)
    const long long R = (long long)max(1, a.super_rounds) * n_blocks * kScanWaves;
    const long long full = a.super_rounds > 0 ? n_tiles / R : 0, rem = n_tiles - full * R;
    const long long s_lo = R * before / total, s_hi = R * (before + mine) / total;
    const long long t_lo = rem * before / total, t_hi = rem * (before + mine) / total;
    wk.wid = (((int)blockIdx.x - my_round * n_cu) >> 3) * kScanWaves + w;
    wk.nwv = group_blocks(grp) * kScanWaves;
    wk.base = (int)s_lo; wk.q = (int)(s_hi - s_lo); wk.R = (int)R;
    wk.full_q = (int)(full * (s_hi - s_lo));
    wk.tail_base = (int)(full * R + t_lo);
    wk.n_seq = wk.full_q + (int)(t_hi - t_lo);
  }
  const long long clock0 = (long long)wall_clock64();
  scan_corpus_tiles<T, DIM, QB, 1, NT>(a.rows, a.n_rows, qreg, tk, a.self0, a.allow, wk);
  if (adapt && threadIdx.x == 0)
    __hip_atomic_fetch_add(tune + grp, (unsigned)((long long)wall_clock64() - clock0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- publish this block's lists, draw a ticket
  const int n_parts = (int)gridDim.x;
  stamp(17 + 6 * (int)blockIdx.x);
  extern __shared__ __align__(16) unsigned char smem_raw[];
  FusedTail<DIM>& t = *reinterpret_cast<FusedTail<DIM>*>(smem_raw + fused_merge_bytes(QB, kk));
  if constexpr (QB == 1) {
    // one query: the four wave lists merge pairwise (two merges deep instead of three in a row), and wave 0 publishes
    // the result with WRITE-THROUGH (agent-scope, sc1) stores, which need no release fence (buffer_wbl2: ~1.7 us)
    block_tree_merge<kScanWaves>(tk[0], t.ms, t.mi);
    if (w == 0 && lane < kk) {
      __hip_atomic_store(a.part_s + (size_t)blockIdx.x * a.kks + lane, tk[0].s[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.part_i + (size_t)blockIdx.x * a.kks + lane, tk[0].id[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else {
    block_merge_store<QB, 1>(tk, kk, a.part_s, a.part_i, n_parts, blockIdx.x, a.kks);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stamp(18 + 6 * (int)blockIdx.x);
  if (threadIdx.x == 0) {
    if constexpr (QB != 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // plain list stores: written back before the ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned got = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = got == (unsigned)(n_parts - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    t.flag = last;
#pragma unroll
    for (int q = 0; q < QB; ++q) t.inv[q] = inv[q];
  }
  __syncthreads();
  stamp(19 + 6 * (int)blockIdx.x);
  if (!t.flag) return;   // block-uniform: every block but the last arriver is done
  stamp(0);

  // ---- the last arriver: fold, rerank, rank, write — one query after the other
  const int r_need = (kk + 3) >> 2;   // heads per wave that must reach the bound
  for (int q = 0; q < QB; ++q) {
    const float* ps = a.part_s + (size_t)q * n_parts * a.kks;
    const int* pi = a.part_i + (size_t)q * n_parts * a.kks;
    if (threadIdx.x == 0) t.count = 0;
    // the normalised query for the rerank (normalize_rows_kernel's values)
    if (a.rerank)
      for (int c = threadIdx.x; c < DIM; c += kScanThreads) t.q[c] = (float)((double)a.q[q * DIM + c] * t.inv[q]);
    // bound: list l belongs to wave l % 4 (lane (l / 4) % 64, slot l / 256).  The loads of the sweep that follows the
    // bound go out right behind the head loads, so their round trip runs under the bisection.
    unsigned key[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int l = i * 256 + lane * 4 + w;
      const float head = ps[(size_t)min(l, n_parts - 1) * a.kks];   // loads are unconditional (a branch around a load costs a full wait each)
      key[i] = l < n_parts ? score_key(head) : 0u;
    }
    const int n_vec = n_parts * a.kks / 4;
    const f32x4* pv = reinterpret_cast<const f32x4*>(ps);
    f32x4 sv[kFusedSweepAhead];
    auto sweep_load = [&](int v0) {
#pragma unroll
      for (int u = 0; u < kFusedSweepAhead; ++u) {
        const int v = v0 + u * kScanThreads;
        sv[u] = pv[min(v, n_vec - 1)];   // unconditional; vectors past the end are skipped where they are tested
      }
    };
    sweep_load(threadIdx.x);
    unsigned lo = 0u;   // invariant: at least r_need of this wave's heads have key >= lo
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned cand = lo | (1u << bit);
      int c = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) c += __builtin_popcountll(__ballot(key[i] >= cand));
      if (c >= r_need) lo = cand;   // wave-uniform
    }
    if (lane == 0) t.bound[w] = lo;
    __syncthreads();
    if (q == 0) stamp(1);
    const unsigned bkey = min(min(t.bound[0], t.bound[1]), min(t.bound[2], t.bound[3]));
    // fewer than r_need heads in some wave (tiny stores): key 0 -> everything passes
    const float bound = bkey == 0u ? -INFINITY : key_score(bkey);

    // sweep: every candidate that reaches the bound goes to LDS as (score, flat position).  A 16-byte vector never
    // straddles two lists (their stride is a multiple of 4) and a list is sorted, so a vector's FIRST score tells
    // whether any of its four can pass: one compare per vector, the per-element work only behind it (testing all 80
    // scores of a thread one by one cost 3.8 us of VALU time).  Lists whose stride is padded (k' not a multiple of 4)
    // hold garbage behind their k' entries — never in a vector's first place — which is tested away there.
    const bool padded = a.kks != kk;
    for (int v0 = threadIdx.x; v0 < n_vec; v0 += kFusedSweepAhead * kScanThreads) {
      if (v0 != (int)threadIdx.x) sweep_load(v0);   // (the first round is in flight already)
#pragma unroll
      for (int u = 0; u < kFusedSweepAhead; ++u) {
        const int v = v0 + u * kScanThreads;
        if (v < n_vec && sv[u].x >= bound) {
          const float sc[4] = {sv[u].x, sv[u].y, sv[u].z, sv[u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (sc[e] >= bound) {
              const int f = v * 4 + e;
              if (padded && (f % a.kks) >= kk) continue;
              const int slot = atomicAdd(&t.count, 1);
              if (slot < kFusedSurvivorCap) { t.sur_s[slot] = sc[e]; t.sur_i[slot] = f; }
            }
          }
        }
      }
    }
    __syncthreads();
    if (q == 0) stamp(2);
    const int m = t.count;
    WaveTopK<1> fk;
    fk.init(kk);
    if (m <= 64) {   // block-uniform; the usual case (about 1.3 k' survivors)
      // one survivor per lane of wave 0; a lane's place in the list = how many survivors beat it, counted against
      // scalar broadcasts of the others (64 x 4 instructions instead of a 27-stage bitonic network of shuffles)
      if (w == 0) {
        const bool valid = lane < m;
        const float cs = valid ? t.sur_s[lane] : -INFINITY;
        int ci = pi[t.sur_i[valid ? lane : 0]];
        if (!valid) ci = kIdxSentinel;
        if (lane < kk) { t.ls[lane] = -INFINITY; t.li[lane] = kIdxSentinel; }
        int rank = 0;
        for (int j = 0; j < m; ++j) {
          const float sj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cs), j));
          const int ij = __builtin_amdgcn_readlane(ci, j);
          rank += (ij != kIdxSentinel && cand_better(sj, ij, cs, ci)) ? 1 : 0;
        }
        if (ci != kIdxSentinel && rank < kk) { t.ls[rank] = cs; t.li[rank] = ci; }
      }
    } else if (m <= kFusedSurvivorCap) {
      // row indices of the survivors (one round of loads), then bitonic wave sorts of 64 at a time
      for (int s0 = w * 64; s0 < m; s0 += kScanThreads) {
        const int s = s0 + lane;
        const bool valid = s < m;
        const float cs = valid ? t.sur_s[s] : -INFINITY;
        int ci = valid ? pi[t.sur_i[s]] : kIdxSentinel;
        fk.push_lanes_bulk(cs, ci, valid && ci != kIdxSentinel);
      }
    } else {
      fold_sorted_lists(fk, ps, pi, w, n_parts, kScanWaves, kk, a.kks);
    }
    if (q == 0) stamp(3);
    if (m > 64) {   // block-uniform
      block_tree_merge<kScanWaves>(fk, t.ms, t.mi);
      if (w == 0 && lane < kk) { t.ls[lane] = fk.s[0]; t.li[lane] = fk.id[0]; }
    }
    __syncthreads();
    if (q == 0) { stamp(4); if (a.stamps && threadIdx.x == 0) a.stamps[8] = m; }
    if (a.rerank) {
      if (a.rows_f32) fused_rerank<DIM, true>(t.ls, t.li, t.q, kk, a.rows_f32, a.rows_b16);   // (the branch on the pointer stays OUTSIDE the loads)
      else fused_rerank<DIM, false>(t.ls, t.li, t.q, kk, a.rows_f32, a.rows_b16);
      __syncthreads();
    }
    if (q == 0) stamp(5);
    // rank by (distance, row) and write: wave 0 alone, one list entry per lane (k' <= 64).  rank_and_write's form — pad
    // everything, barrier, overwrite — waits for the padding stores at the barrier (~1.5 us); here every output place
    // is written exactly once and no barrier stands between the ranking and the stores.  Same ranks, same bits.
    if (w == 0) {
      const int id = lane < kk ? t.li[lane] : kIdxSentinel;
      const float sc = lane < kk ? t.ls[lane] : 0.f;
      const bool ok = id != kIdxSentinel && sc == sc;
      const float d = 1.0f - sc;
      int rank = 0;
      for (int j = 0; j < kk; ++j) {
        const int oj = t.li[j];
        const float sj = t.ls[j];
        rank += (oj != kIdxSentinel && sj == sj && dist_better(1.0f - sj, oj, d, id)) ? 1 : 0;
      }
      const int n_ok = __builtin_popcountll(__ballot(ok));
      long long* oi = a.out_idx + (size_t)q * a.k;
      float* od = a.out_dist + (size_t)q * a.k;
      if (ok && rank < a.k) { oi[rank] = (long long)id + a.row_offset; od[rank] = d; }
      for (int p = n_ok + lane; p < a.k; p += 64) { oi[p] = -1; od[p] = INFINITY; }   // fewer than k candidates: padding behind them
    }
    __syncthreads();
    if (q == 0) stamp(6);
  }
  if (adapt && w == 0) {
    // the next launch's shares, one group per lane: a group's blocks each did share[g] of work in (sum / blocks) of time;
    // the new share is proportional to that speed (over the block-weighted mean speed), taken half way (one launch's
    // times are noisy) and kept within +-25 % of equal
    const int g = lane;
    const bool live = g < n_groups;
    const unsigned d = live ? __hip_atomic_load(tune + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    const int nb = live ? group_blocks(g) : 0;
    const unsigned shr = live ? a.ticket[kFusedShareOffset + g] : 0u;
    const float sh = shr ? (float)shr : (float)kFusedShareUnit;
    const bool has = nb > 0;
    const bool ok = __ballot(has && d == 0u) == 0ull;   // every group with blocks reported a time
    const float speed = (has && d > 0u) ? sh * (float)nb / (float)d : 0.f;
    const float tot = wave_sum(has ? (float)nb * speed : 0.f), wsum = wave_sum(has ? (float)nb : 0.f);
    if (live) {
      if (ok && has) {
        const float target = (float)kFusedShareUnit * speed / (tot / wsum);
        const float next = fminf(fmaxf(0.5f * (sh + target), 0.75f * kFusedShareUnit), 1.25f * kFusedShareUnit);
        a.ticket[kFusedShareOffset + g] = (unsigned)(next + 0.5f);
      }
      __hip_atomic_store(tune + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // every block has arrived: the ticket goes back to zero for the next launch
  if (threadIdx.x == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace vdbk
