// scan_mfma.h — batched (B >= 16) bf16 path: Q.D^T on CDNA4 MFMA with a fused,
// exact top-k' epilogue (the B x N score matrix never leaves the CU).
// Replaces N sequential batch-1 collection.query calls of the reference's
// evaluation loops (analysis/bidirectional_similarity.py:246-271) with one
// contraction; candidate scores are bf16 x bf16 -> fp32, final scores come from
// the fp32 rerank (select.h).
//
// Tiling: a 512-thread block (8 waves, 4(M) x 2(N)) owns BN=128 queries and a
// contiguous range of the corpus, walked in tiles of BM=256 rows; K=768 in 12
// steps of BK=64.  Both operands are K-contiguous, so A (corpus) and B (query)
// fragments of v_mfma_f32_32x32x16_bf16 are plain 16-byte LDS reads.
//
// Staging: global_load_lds_dwordx4 (no VGPR round trip) into a ring of THREE
// 48 KiB LDS stages; two stages stay in flight across the per-step barrier
// (counted `s_waitcnt vmcnt(6)` + raw s_barrier — a __syncthreads() would drain
// the queue), which is what hides the L2/HBM latency at one block per CU.  The
// 16-byte chunk index is XOR-swizzled with (row>>1)&7 on the *source* address
// so ds_read_b128 is bank-conflict free on 128-byte rows.
//
// C layout (orientation chosen for the epilogue): column = query (lane&31),
// rows = 16 corpus rows per lane -> ONE running threshold per lane per N-tile.
// Epilogue per corpus tile: scores >= the query's running k'-th best are
// appended to an LDS queue (wave-aggregated atomics); after a barrier each wave
// drains the entries of the 16 queries it owns into sorted per-query lists kept
// in its REGISTERS (one entry per lane, k' <= 64).  Queue overflow (threshold
// warm-up, adversarial order) is handled exactly by re-running the filter from
// the first unpushed accumulator with the raised thresholds.
#pragma once
#include "wave_prims.h"

namespace vdbk {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kMfBM = 256;          // corpus rows per tile
constexpr int kMfBN = 128;          // queries per block
constexpr int kMfBK = 64;           // K per step (128 bytes per row)
constexpr int kMfDim = 768;
constexpr int kMfRowBytes = kMfDim * 2;
constexpr int kMfSteps = kMfDim / kMfBK;
constexpr int kMfThreads = 512;
constexpr int kMfStageA = kMfBM * kMfBK * 2;    // 32 KiB
constexpr int kMfStageB = kMfBN * kMfBK * 2;    // 16 KiB
constexpr int kMfStage = kMfStageA + kMfStageB;
constexpr int kMfNumStages = 3;
constexpr int kMfQueueCap = 1024;
constexpr int kMfCntStride = 32;    // ints between two queries' append counters: one 128-byte line each.  Packed, 16-32 counters
                                    // share a line and the L2 serialises the returning atomics on it (~8 ns each: the
                                    // 1.5 %-hit-rate sample pass of 16 queries spent 0.4 ms on 48k of them)
constexpr int kMfMaxKK = 64;        // register lists: one entry per lane
constexpr int kMfOwnQ = kMfBN / 8;  // queries owned per wave (16)

__host__ __device__ constexpr size_t mf_smem_bytes() {
  return (size_t)kMfNumStages * kMfStage + (size_t)kMfQueueCap * 12 + 64 + (size_t)kMfBN * 4;
}

#ifdef VDB_MFMA_HOST_TU   // kernels of the orchestrating TU (k_mfma.hip) only: every pass-kernel TU includes this header
// fp32 normalised queries [b][768] -> bf16 [b_pad][768] (rows >= b zero); the first kernel of a batched search, so it
// also clears the batch's overflow flag (one launch less than a memset)
__global__ void queries_to_bf16_kernel(const float* __restrict__ qn, int b, int b_pad, uint16_t* __restrict__ out, int* __restrict__ clear_flag) {
  if (clear_flag && blockIdx.x == 0 && threadIdx.x == 0) *clear_flag = 0;
  const long long total = (long long)b_pad * kMfDim;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / kMfDim);
    float v = r < b ? qn[i] : 0.f;
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) u = (u >> 16) | 0x40; else u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    out[i] = (uint16_t)u;
  }
}
#endif

// Which corpus tile is pass-local tile i?
//   skip_n == 0 : i * stride — a strided sample (stride > 1) or every tile (stride 1).
//   skip_n  > 0 : the tiles the sample {0, stride, ..., (skip_n-1)*stride} left out, ascending (stride >= 2):
//                 the last pass then neither re-multiplies the sampled tiles nor loses their candidates
//                 (select_kth_kernel is seeded with the sample's sorted list).
__device__ __forceinline__ int pass_tile(int i, int stride, int skip_n) {
  if (skip_n == 0) return i * stride;
  const int per = stride - 1, first = skip_n * per;
  return i < first ? (i / per) * stride + 1 + i % per : skip_n * stride + (i - first);
}
__device__ __forceinline__ int pass_tile_next(int g, int stride, int skip_n) {
  if (skip_n == 0) return g + stride;
  const int n = g + 1;
  return (n % stride == 0 && n / stride < skip_n) ? n + 1 : n;
}

__device__ __forceinline__ void mf_glds16(const unsigned char* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// per-thread constant source offsets of its 4 corpus + 2 query chunks within a tile
struct MfSrc {
  int a_row[4];   // tile-local row of chunk t
  int a_off[4];   // a_row*1536 + swizzled chunk*16
  int b_off[2];
};

__device__ __forceinline__ MfSrc mf_make_src() {
  MfSrc s;
  const int tid = threadIdx.x;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int L = t * kMfThreads + tid, r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
    s.a_row[t] = r;
    s.a_off[t] = r * kMfRowBytes + c * 16;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int L = t * kMfThreads + tid, r = L >> 3, c = (L & 7) ^ ((r >> 1) & 7);
    s.b_off[t] = r * kMfRowBytes + c * 16;
  }
  return s;
}

// Stage K-step ks of the corpus tile starting at row0 and of the block's queries.
// LDS image per stage: [A 256 rows][B 128 rows], 128 B per row, global chunk c of row r
// at chunk position c ^ ((r>>1)&7).  6 global_load_lds per thread.
__device__ __forceinline__ void mf_issue_stage(const MfSrc& src, const unsigned char* __restrict__ rows, long long row0,
                                               int n_rows, const unsigned char* __restrict__ qblk, int ks,
                                               unsigned char* stage) {
  const int wbase = (threadIdx.x >> 6) * 64;
  const unsigned char* abase = rows + row0 * kMfRowBytes + ks * (kMfBK * 2);
  if (row0 + kMfBM <= n_rows) {          // block-uniform: full tile
#pragma unroll
    for (int t = 0; t < 4; ++t) mf_glds16(abase + src.a_off[t], stage + (t * kMfThreads + wbase) * 16);
  } else {                                // ragged last tile: clamp (masked in the epilogue)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const long long over = row0 + src.a_row[t] - (n_rows - 1);
      const unsigned char* p = abase + src.a_off[t] - (over > 0 ? over * kMfRowBytes : 0);
      mf_glds16(p, stage + (t * kMfThreads + wbase) * 16);
    }
  }
  const unsigned char* bbase = qblk + ks * (kMfBK * 2);
#pragma unroll
  for (int t = 0; t < 2; ++t) mf_glds16(bbase + src.b_off[t], stage + kMfStageA + (t * kMfThreads + wbase) * 16);
}

// one of the 6 pieces of a stage (t = 0..3 corpus, 4..5 queries); same addresses as mf_issue_stage
template <int T>
__device__ __forceinline__ void mf_issue_piece(const MfSrc& src, const unsigned char* __restrict__ rows, long long row0,
                                               int n_rows, const unsigned char* __restrict__ qblk, int ks,
                                               unsigned char* stage) {
  const int wbase = (threadIdx.x >> 6) * 64;
  if constexpr (T < 4) {
    const unsigned char* abase = rows + row0 * kMfRowBytes + ks * (kMfBK * 2);
    const long long over = row0 + src.a_row[T] - (n_rows - 1);
    mf_glds16(abase + src.a_off[T] - (over > 0 ? over * kMfRowBytes : 0), stage + (T * kMfThreads + wbase) * 16);
  } else {
    mf_glds16(qblk + ks * (kMfBK * 2) + src.b_off[T - 4], stage + kMfStageA + ((T - 4) * kMfThreads + wbase) * 16);
  }
}

#ifdef VDB_MFMA_HOST_TU   // kernels of the orchestrating TU (k_mfma.hip) only: every pass-kernel TU includes this header
// insert (cs, ci) into the register list of local query j (compile-time after the switch)
#define VDB_MF_INSERT_CASE(j)                                              \
  case j: {                                                                \
    WaveTopK<1> tk;                                                        \
    tk.kk = kk; tk.s[0] = ls[j]; tk.id[0] = li[j];                         \
    tk.refresh_thr();                                                      \
    if (tk.accepts(cs, ci)) {                                              \
      tk.insert(cs, ci);                                                   \
      ls[j] = tk.s[0]; li[j] = tk.id[0];                                   \
      if (lane == 0) thr_lds[w * kMfOwnQ + j] = tk.thr_s;                  \
    }                                                                      \
  } break;

// grid = n_qtiles * n_splits blocks; block -> (query tile qt, corpus split sp).
//   rows      : [n_rows][768] bf16 corpus (normalised)
//   qb16      : [n_qtiles*128][768] bf16 queries (zero padded)
//   part_*    : [b][n_splits][kk] partial lists (sorted, sentinel padded)
//   self0 >= 0: query q must not return row self0+q
__global__ __launch_bounds__(kMfThreads) void scan_mfma_topk_kernel(const unsigned char* __restrict__ rows, int n_rows,
                                                                     const unsigned char* __restrict__ qb16, int b,
                                                                     int n_qtiles, int n_splits, int tiles_per_split,
                                                                     int kk, int self0, int xcd_map, int ablate,
                                                                     const int* __restrict__ run_flag,
                                                                     float* __restrict__ part_s,
                                                                     int* __restrict__ part_i) {
  if (run_flag && *run_flag == 0) return;   // overflow fallback: runs only when the append path flagged it
  extern __shared__ __align__(16) unsigned char smem[];   // the ONLY LDS object (keeps hipcc from draining vmcnt)
  unsigned char* stages = smem;
  float* q_s = reinterpret_cast<float*>(smem + kMfNumStages * kMfStage);
  int* q_i = reinterpret_cast<int*>(q_s + kMfQueueCap);
  int* q_q = q_i + kMfQueueCap;
  int* ctrl = q_q + kMfQueueCap;            // [0] queue count, [1] overflow flag
  float* thr_lds = reinterpret_cast<float*>(ctrl + 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;        // 4 x 2 waves: 64 corpus rows x 64 queries each
  const int hsel = lane >> 5;

  // block -> (qt, sp); with xcd_map the n_qtiles blocks of one split share an XCD (blockIdx % 8) and its L2
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
  const long long n_tiles = ((long long)n_rows + kMfBM - 1) / kMfBM;
  const long long tile0 = (long long)sp * tiles_per_split;
  const long long tile1 = tile0 + tiles_per_split < n_tiles ? tile0 + tiles_per_split : n_tiles;
  const int my_tiles = tile1 > tile0 ? (int)(tile1 - tile0) : 0;
  const int total_steps = my_tiles * kMfSteps;

  // per-wave register lists of the 16 queries this wave owns: entry `lane` of each
  float ls[kMfOwnQ];
  int li[kMfOwnQ];
#pragma unroll
  for (int j = 0; j < kMfOwnQ; ++j) { ls[j] = -INFINITY; li[j] = kIdxSentinel; }
  for (int i = tid; i < kMfBN; i += kMfThreads) thr_lds[i] = -INFINITY;
  float thr[2] = {-INFINITY, -INFINITY};

  const unsigned char* qblk = qb16 + (long long)qbase * kMfRowBytes;
  const MfSrc src = mf_make_src();
  f32x16_t acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  if (total_steps > 0) mf_issue_stage(src, rows, tile0 * kMfBM, n_rows, qblk, 0, stages);
  if (total_steps > 1) mf_issue_stage(src, rows, tile0 * kMfBM, n_rows, qblk, 1, stages + kMfStage);

  // fragment read offsets (swizzled); per K sub-step s the chunk is 2s + hsel
  const int ar0 = wm * 64 + (lane & 31), ar1 = ar0 + 32;
  const int br0 = wn * 64 + (lane & 31), br1 = br0 + 32;
  const int a0b = ar0 * 128, a0x = (ar0 >> 1) & 7, a1b = ar1 * 128, a1x = (ar1 >> 1) & 7;
  const int b0b = br0 * 128, b0x = (br0 >> 1) & 7, b1b = br1 * 128, b1x = (br1 >> 1) & 7;

  int buf = 0;  // step % 3
  for (int step = 0; step < total_steps; ++step) {
    // stage `step` has landed once at most the newest stage (6 loads) is outstanding
    if (step + 1 < total_steps && !(ablate & 1)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // all waves' pieces landed; everyone is done reading buffer (step-1)%3
    asm volatile("" ::: "memory");    // keep the LDS reads below the barrier (no fence: it would drain vmcnt)
    const bool do_issue = step + 2 < total_steps && !(ablate & 1);
    const int nstep = step + 2;
    const int nbuf = buf == 0 ? 2 : buf - 1;   // (step+2)%3
    const long long nrow0 = (tile0 + nstep / kMfSteps) * kMfBM;
    const int nks = nstep % kMfSteps;
    unsigned char* nstage = stages + nbuf * kMfStage;
    const bool late = (ablate & 8) && w >= 4;          // experiment: waves 4-7 load after their MFMAs
    const bool inter = (ablate & 16) != 0;             // experiment: spread the 6 loads between the MFMA groups
    if (do_issue && !late && !inter) mf_issue_stage(src, rows, nrow0, n_rows, qblk, nks, nstage);
    const unsigned char* sa = stages + buf * kMfStage;
    const unsigned char* sb = sa + kMfStageA;
    if (!(ablate & 2))
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + hsel;
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(sa + a0b + ((c ^ a0x) << 4));
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(sa + a1b + ((c ^ a1x) << 4));
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(sb + b0b + ((c ^ b0x) << 4));
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(sb + b1b + ((c ^ b1x) << 4));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      if (do_issue && inter) {
        if (s == 0) { mf_issue_piece<0>(src, rows, nrow0, n_rows, qblk, nks, nstage); mf_issue_piece<1>(src, rows, nrow0, n_rows, qblk, nks, nstage); }
        if (s == 1) { mf_issue_piece<2>(src, rows, nrow0, n_rows, qblk, nks, nstage); mf_issue_piece<3>(src, rows, nrow0, n_rows, qblk, nks, nstage); }
        if (s == 2) mf_issue_piece<4>(src, rows, nrow0, n_rows, qblk, nks, nstage);
        if (s == 3) mf_issue_piece<5>(src, rows, nrow0, n_rows, qblk, nks, nstage);
      }
    }
    if (do_issue && late && !inter) mf_issue_stage(src, rows, nrow0, n_rows, qblk, nks, nstage);
    buf = buf == 2 ? 0 : buf + 1;
    if (step % kMfSteps != kMfSteps - 1 || (ablate & 4)) continue;   // ablate: diagnostic timing builds only

    // ------------------------- fused top-k' epilogue -------------------------
    const long long tile_row0 = (tile0 + step / kMfSteps) * kMfBM;
    const bool needs_mask = (tile_row0 + kMfBM > n_rows) || (qbase + kMfBN > b) || (self0 >= 0);  // block-uniform
    int r_start = 0;
    while (true) {  // block-uniform
      if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
      for (int i = tid; i < kMfQueueCap; i += kMfThreads) q_q[i] = -1;
      __syncthreads();
      bool stopped = false;
#pragma unroll
      for (int idx = 0; idx < 64; ++idx) {
        const int mt = idx >> 5, nt = (idx >> 4) & 1, reg = idx & 15;
        if (stopped || idx < r_start) continue;   // wave-uniform
        const float v = acc[mt][nt][reg];
        bool pass = v >= thr[nt];
        if (__ballot(pass) == 0ull) continue;      // the common case: nothing beats the thresholds
        const int row = (int)tile_row0 + wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hsel;
        const int ql = wn * 64 + nt * 32 + (lane & 31);
        if (needs_mask)
          pass = pass && (row < n_rows) && (qbase + ql < b) && !(self0 >= 0 && row == self0 + qbase + ql);
        const unsigned long long m = __ballot(pass);
        if (m == 0ull) continue;
        const int cnt = __popcll(m);
        int base = 0;
        if (lane == 0) base = atomicAdd(&ctrl[0], cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + cnt > kMfQueueCap) {
          if (lane == 0) ctrl[1] = 1;
          stopped = true;
          r_start = idx;
        } else if (pass) {
          const int p = base + __popcll(m & ((1ull << lane) - 1ull));
          q_s[p] = v; q_i[p] = row; q_q[p] = ql;
        }
      }
      if (!stopped) r_start = 64;
      __syncthreads();
      const int nq = ctrl[0] < kMfQueueCap ? ctrl[0] : kMfQueueCap;
      const int ovf = ctrl[1];
      if (nq > 0) {
        // drain: wave w owns local queries [16w, 16w+16)
        for (int base = 0; base < nq; base += 64) {
          const int e = base + lane;
          const int qv = e < nq ? q_q[e] : -1;
          const float sv = e < nq ? q_s[e] : 0.f;
          const int iv = e < nq ? q_i[e] : 0;
          unsigned long long m = __ballot(qv >= 0 && (qv >> 4) == w);
          while (m != 0ull) {
            const int l = __builtin_ctzll(m);
            m &= m - 1ull;
            const int q = __builtin_amdgcn_readlane(qv, l);
            const float cs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), l));
            const int ci = __builtin_amdgcn_readlane(iv, l);
            switch (q & (kMfOwnQ - 1)) {
              VDB_MF_INSERT_CASE(0) VDB_MF_INSERT_CASE(1) VDB_MF_INSERT_CASE(2) VDB_MF_INSERT_CASE(3)
              VDB_MF_INSERT_CASE(4) VDB_MF_INSERT_CASE(5) VDB_MF_INSERT_CASE(6) VDB_MF_INSERT_CASE(7)
              VDB_MF_INSERT_CASE(8) VDB_MF_INSERT_CASE(9) VDB_MF_INSERT_CASE(10) VDB_MF_INSERT_CASE(11)
              VDB_MF_INSERT_CASE(12) VDB_MF_INSERT_CASE(13) VDB_MF_INSERT_CASE(14) VDB_MF_INSERT_CASE(15)
            }
          }
        }
        __syncthreads();
        thr[0] = thr_lds[wn * 64 + (lane & 31)];
        thr[1] = thr_lds[wn * 64 + 32 + (lane & 31)];
      }
      if (!ovf) break;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  }

  // each wave writes the lists it owns: part[q][sp][0..kk)
#pragma unroll
  for (int j = 0; j < kMfOwnQ; ++j) {
    const int q = qbase + w * kMfOwnQ + j;
    if (q < b && lane < kk) {
      const size_t o = ((size_t)q * n_splits + sp) * kk + lane;
      part_s[o] = ls[j];
      part_i[o] = li[j];
    }
  }
}
#undef VDB_MF_INSERT_CASE
#endif

inline bool mfma_path_supported(int kk) { return kk <= kMfMaxKK; }

}  // namespace vdbk
