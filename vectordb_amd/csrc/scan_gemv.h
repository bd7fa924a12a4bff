// scan_gemv.h — the HBM-bound hot kernel: stream the corpus once, score every
// row against QB queries held in registers, keep a fused per-wave top-k.
// Replaces chroma-hnswlib's searchKnn behind collection.query
// (reference: src/vectordb.py:259-264).  The score matrix is never written.
//
// Data layout: rows are row-major, DIM elements, 16-byte aligned, L2-normalised.
// A wave walks the corpus as a flat array of 16-byte chunks: one
// global_load_dwordx4 per lane = 1 KiB per wave-instruction, fully coalesced.
// L loads cover R whole rows (L*64 chunks = R*CPR chunks); U such groups are
// issued back to back (12 KiB in flight per wave at DIM=768).
#pragma once
#include "wave_prims.h"

namespace vdbk {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kScanThreads = 256;
#ifndef VDB_SCAN_LOADS_IN_FLIGHT
#define VDB_SCAN_LOADS_IN_FLIGHT 12
#endif
constexpr int kScanLoadsInFlight = VDB_SCAN_LOADS_IN_FLIGHT;   // 16-byte loads a lane keeps in flight (1 KiB per wave each)
constexpr int kScanWaves = kScanThreads / 64;

__host__ __device__ constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int kPerChunk = 4; };
struct bf16_t { uint16_t bits; };
template <> struct ElemTraits<bf16_t> { static constexpr int kPerChunk = 8; };

template <bool NT>
__device__ __forceinline__ u32x4 load_chunk(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// dot of one 16-byte chunk with the matching query slice
template <typename T>
__device__ __forceinline__ float chunk_dot(const u32x4 v, const float* q);
template <>
__device__ __forceinline__ float chunk_dot<float>(const u32x4 v, const float* q) {
  float a = __uint_as_float(v.x) * q[0];
  a = fmaf(__uint_as_float(v.y), q[1], a);
  a = fmaf(__uint_as_float(v.z), q[2], a);
  a = fmaf(__uint_as_float(v.w), q[3], a);
  return a;
}
template <>
__device__ __forceinline__ float chunk_dot<bf16_t>(const u32x4 v, const float* q) {
  float a = __uint_as_float(v.x << 16) * q[0];
  a = fmaf(__uint_as_float(v.x & 0xffff0000u), q[1], a);
  a = fmaf(__uint_as_float(v.y << 16), q[2], a);
  a = fmaf(__uint_as_float(v.y & 0xffff0000u), q[3], a);
  a = fmaf(__uint_as_float(v.z << 16), q[4], a);
  a = fmaf(__uint_as_float(v.z & 0xffff0000u), q[5], a);
  a = fmaf(__uint_as_float(v.w << 16), q[6], a);
  a = fmaf(__uint_as_float(v.w & 0xffff0000u), q[7], a);
  return a;
}

// Block-level merge of the per-wave lists + write of one partial list per query:
// out layout [q][part][kk].
template <int QB, int E>
__device__ __forceinline__ void block_merge_store(WaveTopK<E> (&tk)[QB], int kk, float* __restrict__ part_s,
                                                  int* __restrict__ part_i, int n_parts, int part, int stride = 0) {
  if (stride == 0) stride = kk;   // floats between two lists of the output (>= kk)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* ls = reinterpret_cast<float*>(smem_raw);          // [waves][QB][kk]
  int* li = reinterpret_cast<int*>(ls + kScanWaves * QB * kk);
  const int w = threadIdx.x >> 6;
  const int lane = lane_id();
#pragma unroll
  for (int q = 0; q < QB; ++q) tk[q].store(ls + (w * QB + q) * kk, li + (w * QB + q) * kk);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < QB; ++q) {
    if ((q % kScanWaves) == w) {  // wave w finishes query q (wave-uniform)
      for (int ow = 0; ow < kScanWaves; ++ow) {
        if (ow == w) continue;
        const float* os = ls + (ow * QB + q) * kk;
        const int* oi = li + (ow * QB + q) * kk;
        if constexpr (E == 1) {
          const bool valid = lane < kk;   // the other wave's list is sorted: bitonic merge
          tk[q].merge_sorted(valid ? os[lane] : -INFINITY, valid ? oi[lane] : kIdxSentinel);
        } else {
          for (int base = 0; base < kk; base += 64) {
            const int p = base + lane;
            const bool valid = p < kk;
            const float cs = valid ? os[p] : -INFINITY;
            const int ci = valid ? oi[p] : kIdxSentinel;
            tk[q].push_lanes(cs, ci, valid);
          }
        }
      }
      tk[q].store(part_s + ((size_t)q * n_parts + part) * stride, part_i + ((size_t)q * n_parts + part) * stride);
    }
  }
}

// ---------------------------------------------------------------------------
// Fast path: compile-time DIM (768, 384), QB queries per corpus pass.
//   qn      : QB x DIM normalised fp32 queries (device)
//   self0   : >=0 -> query q must not return row self0+q (self-join)
//   allow   : nullptr, or the eligibility bitmap (bit r set = row r may be returned);
//             a tile's rows lie in one 32-bit word, fetched with the tile's loads
//   part_*  : [QB][gridDim.x][kk] partial lists
// ---------------------------------------------------------------------------
// Geometry of the fast path: a wave-load group covers R whole rows with L loads; U groups make a tile.
template <typename T, int DIM>
struct ScanGeom {
  static constexpr int EPC = ElemTraits<T>::kPerChunk;   // elements per 16-byte chunk
  static constexpr int CPR = DIM / EPC;                  // chunks per row
  static constexpr int G = cgcd(CPR, 64);
  static constexpr int L = CPR / G;                      // wave-loads per group
  static constexpr int R = 64 / G;                       // rows per group
  static constexpr int U = (L >= kScanLoadsInFlight) ? 1 : (kScanLoadsInFlight / L);     // groups per tile
  static constexpr int RPT = U * R;                      // rows per tile
  static_assert(DIM % EPC == 0, "row must be a whole number of 16-byte chunks");
  static_assert(RPT <= 32 && 32 % RPT == 0, "a tile's eligibility bits must lie in one word");
};

// One tile (RPT rows) of the corpus pass: all its loads at once, then the dots, the wave reductions and the insertions.
template <typename T, int DIM, int QB, int E, bool NT>
__device__ __forceinline__ void scan_one_tile(const u32x4* __restrict__ rows, int n_rows, long long total_chunks, long long tile, unsigned elig,
                                              const float (&qreg)[QB][ScanGeom<T, DIM>::L][ScanGeom<T, DIM>::EPC],
                                              WaveTopK<E> (&tk)[QB], int self0) {
  using GM = ScanGeom<T, DIM>;
  constexpr int CPR = GM::CPR, L = GM::L, R = GM::R, U = GM::U, RPT = GM::RPT;
  const int lane = lane_id();
  const long long c0 = tile * (long long)(RPT * CPR) + lane;
  const long long row0 = tile * RPT;
  u32x4 v[U * L];
  if ((tile + 1) * RPT <= (long long)n_rows) {
#pragma unroll
    for (int t = 0; t < U * L; ++t) v[t] = load_chunk<NT>(rows + c0 + t * 64);
  } else {
#pragma unroll
    for (int t = 0; t < U * L; ++t) {
      const long long c = c0 + t * 64;
      v[t] = (c < total_chunks) ? load_chunk<NT>(rows + c) : u32x4{0u, 0u, 0u, 0u};
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int q = 0; q < QB; ++q) {
      float p[L];
#pragma unroll
      for (int j = 0; j < L; ++j) p[j] = chunk_dot<T>(v[u * L + j], qreg[q][j]);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < L; ++j) {
          // lanes of load j that belong to row r of the group: [lo, hi)
          const int lo = r * CPR - j * 64, hi = lo + CPR;
          if (lo <= 0 && hi >= 64) acc += p[j];
          else if (hi <= 0 || lo >= 64) { /* no lane */ }
          else acc += (lane >= lo && lane < hi) ? p[j] : 0.f;
        }
        const float tot = wave_sum(acc);
        const long long row = row0 + u * R + r;
        if (row < (long long)n_rows && ((elig >> (u * R + r)) & 1u)) {
          const int ri = (int)row;
          const bool self = (self0 >= 0) && (ri == self0 + q);
          if (!self && tk[q].accepts(tot, ri)) tk[q].insert(tot, ri);
        }
      }
    }
  }
}

// Which tiles a wave visits.  The waves that share a SEQUENCE of tiles (all waves of the grid in the plain scan; the
// waves of one class of blocks in the fused kernel) take its members wid, wid + nwv, ... < n_seq.  Member s of a sequence
// lies in super-round s / q at offset s % q — the class owns the q tiles from `base` on in each of the first
// full_q / q super-rounds of R tiles — and behind those in a last range of its own from tail_base on.
// Why super-rounds: the whole grid then walks the corpus as ONE window (a few tens of MB wide), each class on its own
// slice of it.  Streams that walk ranges of their own far apart (1.2 GB: the 16 class ranges this kernel used first)
// lose ~5 % of the read rate in SOME regions of HBM, any two such streams are enough (benchmarks/placement_probe.hip,
// profiles/r03_placement.txt: 6.7 vs 7.07 TB/s, the same windows of one allocation every time), which made a store's
// scan rate depend on where it was allocated; one window of per-XCD slices reads 7.15-7.19 TB/s in every region.
struct TileWalk {
  int wid, nwv, n_seq;
  int base, q, R, full_q, tail_base;
  __device__ __forceinline__ static TileWalk plain(long long n_tiles, long long wid_, long long nwv_) {
    return TileWalk{(int)wid_, (int)nwv_, (int)n_tiles, 0, 0, 0, 0, 0};
  }
};

// The corpus pass of one wave: every tile of its walk scored against the QB queries in qreg, winners into tk.
template <typename T, int DIM, int QB, int E, bool NT>
__device__ __forceinline__ void scan_corpus_tiles(const void* __restrict__ rows_v, int n_rows,
                                                  const float (&qreg)[QB][ScanGeom<T, DIM>::L][ScanGeom<T, DIM>::EPC],
                                                  WaveTopK<E> (&tk)[QB], int self0, const unsigned* __restrict__ allow,
                                                  const TileWalk wk) {
  using GM = ScanGeom<T, DIM>;
  constexpr int CPR = GM::CPR, RPT = GM::RPT;
  const u32x4* __restrict__ rows = reinterpret_cast<const u32x4*>(rows_v);
  const long long total_chunks = (long long)n_rows * CPR;

  // position of member s inside the super-rounds: round r, offset o (one division per wave, then increments)
  int s = wk.wid, r = 0, o = 0;
  if (wk.full_q > 0) { r = s / wk.q; o = s - r * wk.q; }
  auto tile_of = [&](int s_, int r_, int o_) -> long long {
    return s_ < wk.full_q ? (long long)wk.base + (long long)r_ * wk.R + o_ : (long long)wk.tail_base + (s_ - wk.full_q);
  };

  // eligibility bits of a tile's rows (bit j = row tile*RPT + j), fetched ONE TILE AHEAD so that the (wave-uniform)
  // load hides behind the current tile; a tile with no eligible row is not read at all, so a selective where=
  // filter streams only the tiles it needs (1 % eligible at random: 92 % of the 8-row tiles are skipped)
  constexpr unsigned kTileBits = RPT == 32 ? 0xffffffffu : ((1u << RPT) - 1u);
  auto tile_bits = [&](long long t) -> unsigned {
    const long long r0 = t * RPT;
    return (allow[r0 >> 5] >> (unsigned)(r0 & 31)) & kTileBits;
  };
  long long tile = s < wk.n_seq ? tile_of(s, r, o) : 0;
  unsigned elig_next = (allow && s < wk.n_seq) ? tile_bits(tile) : kTileBits;

  while (s < wk.n_seq) {
    const long long here = tile;
    const unsigned elig = elig_next;
    s += wk.nwv;
    if (wk.full_q > 0) { o += wk.nwv; while (o >= wk.q) { o -= wk.q; ++r; } }
    if (s < wk.n_seq) {
      tile = tile_of(s, r, o);
      if (allow) elig_next = tile_bits(tile);
    }
    if (elig == 0u) continue;
    scan_one_tile<T, DIM, QB, E, NT>(rows, n_rows, total_chunks, here, elig, qreg, tk, self0);
  }
}

template <typename T, int DIM, int QB, int E, bool NT>
__global__ __launch_bounds__(kScanThreads) void scan_topk_kernel(const void* __restrict__ rows_v, int n_rows,
                                                                  const float* __restrict__ qn, int kk,
                                                                  int self0, const unsigned* __restrict__ allow,
                                                                  float* __restrict__ part_s,
                                                                  int* __restrict__ part_i, int xcd_slices) {
  using GM = ScanGeom<T, DIM>;
  constexpr int EPC = GM::EPC, CPR = GM::CPR, L = GM::L;
  const int lane = lane_id();
  // blockIdx.y = which group of QB queries (a batch against a small store goes out as ONE launch of
  // gridDim.y groups instead of gridDim.y launches; lists land where those launches would put them)
  {
    const size_t g = blockIdx.y;
    qn += g * QB * DIM;
    part_s += g * QB * gridDim.x * kk;
    part_i += g * QB * gridDim.x * kk;
    if (self0 >= 0) self0 += (int)g * QB;
  }

  // per-lane query slices: load j of a group covers chunk (j*64+lane) % CPR of its row
  float qreg[QB][L][EPC];
#pragma unroll
  for (int q = 0; q < QB; ++q)
#pragma unroll
    for (int j = 0; j < L; ++j) {
      const int cc = (j * 64 + lane) % CPR;
#pragma unroll
      for (int e = 0; e < EPC; ++e) qreg[q][j][e] = qn[q * DIM + cc * EPC + e];
    }

  WaveTopK<E> tk[QB];
#pragma unroll
  for (int q = 0; q < QB; ++q) tk[q].init(kk);

  constexpr int RPT = GM::RPT;
  // blocks b, b + 8, ... share an XCD: they take ADJACENT tiles, so that each grid-wide step over the corpus is eight
  // contiguous per-XCD slices of one window (profiles/r03_placement.txt: 7.15-7.19 against 7.06-7.09 TB/s)
  const int blk = (xcd_slices && (gridDim.x & 7u) == 0u) ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  scan_corpus_tiles<T, DIM, QB, E, NT>(rows_v, n_rows, qreg, tk, self0, allow,
                                       TileWalk::plain(((long long)n_rows + RPT - 1) / RPT, (long long)blk * kScanWaves + (threadIdx.x >> 6),
                                                       (long long)gridDim.x * kScanWaves));
  block_merge_store<QB, E>(tk, kk, part_s, part_i, gridDim.x, blockIdx.x);
}

// ---------------------------------------------------------------------------
// Generic path: any (padded) row length ld (multiple of the chunk size), one
// row per wave step, query staged in LDS.  One query per pass.
// ---------------------------------------------------------------------------
template <typename T, int E, bool NT>
__global__ __launch_bounds__(kScanThreads) void scan_topk_generic_kernel(const void* __restrict__ rows_v, int n_rows,
                                                                         int ld, const float* __restrict__ qn,
                                                                         int kk, int self_row,
                                                                         const unsigned* __restrict__ allow,
                                                                         float* __restrict__ part_s,
                                                                         int* __restrict__ part_i) {
  constexpr int EPC = ElemTraits<T>::kPerChunk;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // query lives after the merge area: [waves*kk floats][waves*kk ints][ld floats]
  float* qs = reinterpret_cast<float*>(smem_raw) + 2 * kScanWaves * kk;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) qs[c] = qn[c];
  __syncthreads();

  const u32x4* __restrict__ rows = reinterpret_cast<const u32x4*>(rows_v);
  const int cpr = ld / EPC;
  const int lane = lane_id();
  const int w = threadIdx.x >> 6;
  WaveTopK<E> tk[1];
  tk[0].init(kk);
  const long long gw = (long long)blockIdx.x * kScanWaves + w;
  const long long nw = (long long)gridDim.x * kScanWaves;
  for (long long row = gw; row < (long long)n_rows; row += nw) {
    if (allow && !((allow[row >> 5] >> (unsigned)(row & 31)) & 1u)) continue;   // wave-uniform
    float acc = 0.f;
    for (int c = lane; c < cpr; c += 64) {
      const u32x4 v = load_chunk<NT>(rows + row * cpr + c);
      acc += chunk_dot<T>(v, qs + c * EPC);
    }
    const float tot = wave_sum(acc);
    const int ri = (int)row;
    if (ri != self_row && tk[0].accepts(tot, ri)) tk[0].insert(tot, ri);
  }
  __syncthreads();  // query reads done before the merge area is reused (disjoint, but keep order explicit)
  block_merge_store<1, E>(tk, kk, part_s, part_i, gridDim.x, blockIdx.x);
}

}  // namespace vdbk
