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
// fragments of v_mfma_f32_32x32x16_bf16 are plain 16-byte LDS reads.  Tiles
// are staged global->LDS with global_load_lds_dwordx4 (no VGPR round trip),
// double buffered; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 on
// the *source* address so ds_read_b128 is bank-conflict free on 128-byte rows.
// C layout (orientation chosen for the epilogue): column = query (lane&31),
// rows = 16 corpus rows per lane -> ONE running threshold per lane per N-tile.
//
// Epilogue per corpus tile: scores >= the query's running k'-th best are
// appended to an LDS queue (wave-aggregated atomics); after a barrier each wave
// drains the entries of the 16 queries it owns into sorted per-query lists in
// LDS (register insertion, lists never leave the CU until the block ends).
// Queue overflow (threshold warm-up, adversarial order) is handled exactly by
// re-running the filter from the first unpushed accumulator with the raised
// thresholds.
#pragma once
#include "wave_prims.h"

namespace vdbk {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

constexpr int kMfBM = 256;          // corpus rows per tile
constexpr int kMfBN = 128;          // queries per block
constexpr int kMfBK = 64;           // K per step (128 bytes per row)
constexpr int kMfDim = 768;
constexpr int kMfSteps = kMfDim / kMfBK;
constexpr int kMfThreads = 512;
constexpr int kMfStageA = kMfBM * kMfBK * 2;    // 32 KiB
constexpr int kMfStageB = kMfBN * kMfBK * 2;    // 16 KiB
constexpr int kMfStage = kMfStageA + kMfStageB;
constexpr int kMfQueueCap = 1024;
constexpr int kMfMaxKK = 48;        // LDS budget: 96 K staging + 12 K queue + 128*kk*8 lists

__host__ __device__ constexpr size_t mf_smem_bytes(int kk) {
  return (size_t)2 * kMfStage + (size_t)kMfQueueCap * 12 + 64 + (size_t)kMfBN * 4 + (size_t)kMfBN * kk * 8;
}

// fp32 normalised queries [b][768] -> bf16 [b_pad][768] (rows >= b zero)
__global__ void queries_to_bf16_kernel(const float* __restrict__ qn, int b, int b_pad, uint16_t* __restrict__ out) {
  const long long total = (long long)b_pad * kMfDim;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / kMfDim);
    float v = r < b ? qn[i] : 0.f;
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) u = (u >> 16) | 0x40; else u = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
    out[i] = (uint16_t)u;
  }
}

__device__ __forceinline__ void mf_glds16(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gptr,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Stage K-step `ks` of corpus tile rows [row0, row0+256) and of the block's 128
// queries into one LDS stage.  LDS image: row-major, 128 B per row, chunk c of row
// r stored at chunk position c ^ ((r>>1)&7).
__device__ __forceinline__ void mf_issue_stage(const unsigned char* __restrict__ rows, long long row0, int n_rows,
                                               const unsigned char* __restrict__ qb16, int ks, unsigned char* stage) {
  const int tid = threadIdx.x;
  const int wbase = (tid >> 6) * 64;
#pragma unroll
  for (int t = 0; t < 4; ++t) {       // A: 2048 chunks
    const int L = t * kMfThreads + tid;
    const int r = L >> 3, cp = L & 7;
    const int c = cp ^ ((r >> 1) & 7);
    long long gr = row0 + r;
    if (gr >= n_rows) gr = n_rows - 1;  // clamp: rows past the end are masked in the epilogue
    mf_glds16(rows + gr * (kMfDim * 2) + ks * (kMfBK * 2) + c * 16, stage + (t * kMfThreads + wbase) * 16);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {       // B: 1024 chunks
    const int L = t * kMfThreads + tid;
    const int r = L >> 3, cp = L & 7;
    const int c = cp ^ ((r >> 1) & 7);
    mf_glds16(qb16 + (long long)r * (kMfDim * 2) + ks * (kMfBK * 2) + c * 16,
              stage + kMfStageA + (t * kMfThreads + wbase) * 16);
  }
}

// grid = n_qtiles * n_splits blocks; block -> (query tile qt, corpus split sp).
//   rows      : [n_rows][768] bf16 corpus (normalised)
//   qb16      : [n_qtiles*128][768] bf16 queries (zero padded)
//   part_*    : [b][n_splits][kk] partial lists (sorted, sentinel padded)
//   self0 >= 0: query q must not return row self0+q
__global__ __launch_bounds__(kMfThreads) void scan_mfma_topk_kernel(const unsigned char* __restrict__ rows, int n_rows,
                                                                     const unsigned char* __restrict__ qb16, int b,
                                                                     int n_qtiles, int n_splits, int tiles_per_split,
                                                                     int kk, int self0, int xcd_map,
                                                                     float* __restrict__ part_s,
                                                                     int* __restrict__ part_i) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* stages = smem;
  float* q_s = reinterpret_cast<float*>(smem + 2 * kMfStage);
  int* q_i = reinterpret_cast<int*>(q_s + kMfQueueCap);
  int* q_q = q_i + kMfQueueCap;
  int* ctrl = q_q + kMfQueueCap;            // [0] queue count, [1] overflow flag
  float* thr_lds = reinterpret_cast<float*>(ctrl + 16);
  float* list_s = thr_lds + kMfBN;
  int* list_i = reinterpret_cast<int*>(list_s + kMfBN * kk);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;        // 4 x 2 waves: 64 corpus rows x 64 queries each

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

  for (int i = tid; i < kMfBN * kk; i += kMfThreads) { list_s[i] = -INFINITY; list_i[i] = kIdxSentinel; }
  for (int i = tid; i < kMfBN; i += kMfThreads) thr_lds[i] = -INFINITY;
  float thr[2] = {-INFINITY, -INFINITY};

  const unsigned char* qblk = qb16 + (long long)qbase * (kMfDim * 2);
  f32x16_t acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int total_steps = my_tiles * kMfSteps;
  if (total_steps > 0) mf_issue_stage(rows, tile0 * kMfBM, n_rows, qblk, 0, stages);

  // fragment read addresses (swizzled); per K sub-step s the chunk is 2s + (lane>>5)
  const int ar0 = wm * 64 + (lane & 31), ar1 = ar0 + 32;
  const int br0 = wn * 64 + (lane & 31), br1 = br0 + 32;
  const int hsel = lane >> 5;

  for (int step = 0; step < total_steps; ++step) {
    __syncthreads();   // stage `step` landed (vmcnt(0)) and every wave finished reading the other buffer
    if (step + 1 < total_steps) {
      const int nstep = step + 1;
      mf_issue_stage(rows, (tile0 + nstep / kMfSteps) * kMfBM, n_rows, qblk, nstep % kMfSteps,
                     stages + (nstep & 1) * kMfStage);
    }
    const unsigned char* sa = stages + (step & 1) * kMfStage;
    const unsigned char* sb = sa + kMfStageA;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 2 * s + hsel;
      const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(sa + ar0 * 128 + ((c ^ ((ar0 >> 1) & 7)) << 4));
      const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(sa + ar1 * 128 + ((c ^ ((ar1 >> 1) & 7)) << 4));
      const bf16x8_t b0 = *reinterpret_cast<const bf16x8_t*>(sb + br0 * 128 + ((c ^ ((br0 >> 1) & 7)) << 4));
      const bf16x8_t b1 = *reinterpret_cast<const bf16x8_t*>(sb + br1 * 128 + ((c ^ ((br1 >> 1) & 7)) << 4));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (step % kMfSteps != kMfSteps - 1) continue;

    // ------------------------- fused top-k' epilogue -------------------------
    const long long tile_row0 = (tile0 + step / kMfSteps) * kMfBM;
    int r_start = 0;
    while (true) {  // block-uniform
      if (tid == 0) { ctrl[0] = 0; ctrl[1] = 0; }
      for (int i = tid; i < kMfQueueCap; i += kMfThreads) q_q[i] = -1;
      __syncthreads();
      bool stopped = false;
#pragma unroll
      for (int idx = 0; idx < 64; ++idx) {
        const int mt = idx >> 5, nt = (idx >> 4) & 1, reg = idx & 15;
        if (!stopped && idx >= r_start) {  // wave-uniform
          const float v = acc[mt][nt][reg];
          const long long row = tile_row0 + wm * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hsel;
          const int ql = wn * 64 + nt * 32 + (lane & 31);
          const bool pass = (v >= thr[nt]) && (row < n_rows) && (qbase + ql < b) &&
                            !(self0 >= 0 && row == (long long)self0 + qbase + ql);
          const unsigned long long m = __ballot(pass);
          if (m != 0ull) {
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
              q_s[p] = v; q_i[p] = (int)row; q_q[p] = ql;
            }
          }
        }
      }
      if (!stopped) r_start = 64;
      __syncthreads();
      const int nq = ctrl[0] < kMfQueueCap ? ctrl[0] : kMfQueueCap;
      const int ovf = ctrl[1];
      if (nq > 0) {
        // drain: wave w owns queries [16w, 16w+16)
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
            WaveTopK<1> tk;
            tk.kk = kk;
            tk.s[0] = lane < kk ? list_s[q * kk + lane] : -INFINITY;
            tk.id[0] = lane < kk ? list_i[q * kk + lane] : kIdxSentinel;
            tk.refresh_thr();
            if (tk.accepts(cs, ci)) {
              tk.insert(cs, ci);
              tk.store(list_s + q * kk, list_i + q * kk);
              if (lane == 0) thr_lds[q] = tk.thr_s;
            }
          }
        }
      }
      __syncthreads();
      thr[0] = thr_lds[wn * 64 + (lane & 31)];
      thr[1] = thr_lds[wn * 64 + 32 + (lane & 31)];
      if (!ovf) break;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  }

  __syncthreads();
  for (int i = tid; i < kMfBN * kk; i += kMfThreads) {
    const int q = i / kk, p = i - q * kk;
    if (qbase + q < b) {
      const size_t o = ((size_t)(qbase + q) * n_splits + sp) * kk + p;
      part_s[o] = list_s[i];
      part_i[o] = list_i[i];
    }
  }
}

inline bool mfma_path_supported(int kk) { return kk <= kMfMaxKK; }

}  // namespace vdbk
