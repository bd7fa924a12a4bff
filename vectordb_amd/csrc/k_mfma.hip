// k_mfma.hip — the batched bf16 path: Q.D^T on MFMA with the threshold/append protocol of scan_mfma3.h.
// Host orchestration of one batched search (all buffers are the caller's) + the selection and fallback kernels; the pass
// kernels instantiate in their own translation units behind mfma_pass.h.
#include <algorithm>
#include <cmath>
#include "launch.h"
#include "mfma_pass.h"
#define VDB_MFMA_HOST_TU
#include "scan_mfma3.h"

namespace vdbk {
namespace {

struct MfGrid { int n_splits, xcd_map, tiles_per_split; };
MfGrid mf_grid(int n_cu, int n_qt, long long pass_tiles, int max_splits) {
  MfGrid g;
  g.n_splits = (int)std::max<long long>(1, std::min<long long>(pass_tiles, std::max(1, std::min(max_splits, n_cu / n_qt))));
  g.xcd_map = 0;
  if (g.n_splits >= 8) { g.n_splits &= ~7; g.xcd_map = 1; }   // the n_qt blocks of a split share one XCD's L2
  g.tiles_per_split = (int)((pass_tiles + g.n_splits - 1) / g.n_splits);
  return g;
}

static_assert(kMfDenseLdC == kMf3DenseLd && kMfRowBytesC == kMfRowBytes && kMfCntStrideC == kMfCntStride, "launch.h mirrors the kernels' constants");

// waves per query in select_kth_kernel: 16 (one 1024-thread block per query, two per CU at a time) 11.63 / 1.669 ms per
// 1024-query batch at 10M / 1.25M rows, 8 waves 11.59 / 1.647, 4 waves 11.57 / 1.642 (three libraries alternated on one box)
#ifndef VDB_SEL_WAVES
#define VDB_SEL_WAVES 4
#endif
#define MF_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace

bool mfma_kk_supported(int kk) { return mfma_path_supported(kk); }

// Which pass kernel serves a batch of b queries:
//   b <= 64                         -> 6: the streaming kernel (scan_stream.h): corpus straight from HBM into MFMA
//                                         fragments, no LDS staging of the corpus — HBM-bound like the GEMV scan;
//   a 256-wide query tile pads the batch no more than 1.5x further than a 128-wide one -> 5 (scan_mfma5.h; it runs
//                                         ~1.9x the rate of the 128-wide tiles per query; only b <= 128 fails the test);
//   otherwise                       -> 4: the same kernel on a 128-wide query tile (scan_mfma6.h, NB = 2; it replaced the
//                                         warp-specialised 256x128 kernel of round 1: 2.74 / 2.78 vs 2.90 / 3.15 ms at 65 / 128 queries).
MfPlan mfma_plan(int b, int n_rows, int n_cu, int variant_opt) {
  (void)n_rows;
  MfPlan p;
  p.n_qt_fb = (b + kMfBN - 1) / kMfBN;
  const int pad_wide = ((b + kM5BNC - 1) / kM5BNC) * kM5BNC;
  const bool wide = variant_opt >= 5 && 2 * pad_wide <= 3 * p.n_qt_fb * kMfBN;
  const bool stream = variant_opt >= 6 && b <= kStMaxBC;
  p.variant = stream ? 6 : wide ? 5 : 4;
  p.n_qt = stream ? 1 : wide ? pad_wide / kM5BNC : p.n_qt_fb;
  p.b_pad = std::max(wide ? pad_wide : p.n_qt_fb * kMfBN, p.n_qt_fb * kMfBN);   // >= the fallback's 128-wide tiles either way
  const long long n_tiles = ((long long)n_rows + kMfBM - 1) / kMfBM;
  p.fb_splits = mf_grid(n_cu, p.n_qt_fb, n_tiles, 64).n_splits;                   // fallback grid (<= 64 partial lists per query)
  return p;
}

hipError_t mfma_prepare_device() {
  MF_TRY(hipFuncSetAttribute((const void*)scan_mfma_topk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mf_smem_bytes()));
  MF_TRY(wide16_prepare_device());
  MF_TRY(stream_prepare());
  return hipSuccess;
}

hipError_t mfma_search_launch(const MfSearchArgs& a, int* skipped_tiles) {
  const int b = a.b, kk = a.kk, n_rows = a.n_rows, cap = a.cap, self0 = a.self0;
  const MfPlan pl = mfma_plan(b, n_rows, a.n_cu, a.variant);
  const int variant = pl.variant, n_qt = pl.n_qt, n_qt_fb = pl.n_qt_fb;
  const long long n_tiles = ((long long)n_rows + kMfBM - 1) / kMfBM;
  const unsigned char* rows = a.rows_bf16;
  const unsigned char* qb16 = a.qb16;
  float* cand_s = a.cand_s; int* cand_i = a.cand_i; int* cand_cnt = a.cand_cnt;
  int* ovf = cand_cnt + (size_t)b * kMfCntStride;
  float* thr = a.thr; float* dense = a.dense;
  hipStream_t st = a.st;
  if (skipped_tiles) *skipped_tiles = 0;
  const MfGrid fb = mf_grid(a.n_cu, n_qt_fb, n_tiles, 64);

  if (!a.qb16_ready) {
    hipLaunchKernelGGL(queries_to_bf16_kernel, dim3(std::min(1024, (pl.b_pad * kMfDim + 255) / 256)), dim3(256), 0, st,
                       a.qn, b, pl.b_pad, (uint16_t*)a.qb16, ovf);   // also clears the overflow flag
    MF_TRY(hipGetLastError());
  }

  // one pass of whichever kernel serves this batch; mode 0 = dense sample, 1 = append
  auto launch_pass = [&](int mode, long long pass_tiles, int stride, int skip_n) -> hipError_t {
    const MfGrid g = mf_grid(a.n_cu, n_qt, pass_tiles, 1 << 30);
    MfPassArgs p;
    p.mode = mode; p.rows = rows; p.n_rows = n_rows; p.qb16 = qb16; p.b = b;
    p.n_qt = n_qt; p.n_splits = g.n_splits; p.tiles_per_split = g.tiles_per_split; p.xcd_map = g.xcd_map;
    p.pass_tiles = (int)pass_tiles; p.stride = stride; p.skip_n = skip_n; p.self0 = self0;
    p.thr = thr; p.dense = dense; p.cand_cnt = cand_cnt; p.cand_s = cand_s; p.cand_i = cand_i; p.cap = cap;
    p.ovf = ovf; p.ablate = a.ablate; p.st = st;
    if (variant == 6) return stream_pass(p, a.n_cu, a.stream_nt);
    if (variant == 5) return wide16_launch_pass(p);
    p.narrow = 1;   // variant 4: the 128-wide query tile of the same kernel
    return wide16_launch_pass(p);
  };

  // pass 0: dense scores of a strided 16-tile sample -> t0 = a lower bound of the k'-th best per query
  long long t_cur = std::min<long long>(n_tiles, std::max(1, std::min(a.dense_tiles, kMf3DenseTiles)));
  {
    const int stride = (int)std::max<long long>(1, n_tiles / t_cur);
    MF_TRY(launch_pass(0, t_cur, stride, 0));
    // rows of the dense matrix beyond the sampled tiles were never written: select over t_cur*256 columns only
    // n_tiles == t_cur: the dense pass saw the whole store and the next pass is the last; either way a valid lower bound suffices
    hipLaunchKernelGGL(dense_lower_bound_kernel, dim3(b), dim3(256), 0, st, dense, kMf3DenseLd, (int)t_cur * kMfBM, kk, thr, cand_cnt);   // also clears the append counters
    MF_TRY(hipGetLastError());
  }
  // append passes over geometrically growing strided samples; the growth factor per pass is
  // at most R = cap/(4 k') so that ~cap/4 candidates per query are expected in each
  const double Rmax = std::max<double>(2.0, (double)cap / (4.0 * kk));
  const double span = (double)n_tiles / (double)t_cur;
  int n_pass = span <= 1.0 ? 1 : std::max(1, (int)std::ceil(std::log(span) / std::log(Rmax) - 0.15));   // up to ~2x Rmax per pass
  // one pass means k' x span hits per query in it, two about 2 k' sqrt(span) at the price of a sample pass and a selection
  if (n_pass == 1 && span > (double)a.one_pass_span) n_pass = 2;
  const double growth = std::pow(span, 1.0 / n_pass);
  // Two passes (the usual case): the last one leaves out the tiles the sample pass multiplied and the final
  // selection is seeded with the sample's sorted list instead — the sample is 2 % of the tiles at 10M rows,
  // 6 % at an 8-GPU shard's 1.25M.
  bool seeded = false;
  int seed_stride = 0, seed_n = 0;   // the sample the seed list came from: tiles 0, stride, ..., (n-1)*stride
  for (int pass = 1; pass <= n_pass; ++pass) {
    const bool last = pass == n_pass;
    long long t_next = last ? n_tiles : std::min<long long>(n_tiles, (long long)std::ceil((double)t_cur * growth));
    if (!last && n_pass == 2) t_next = std::min<long long>(n_tiles, (long long)std::ceil((double)t_cur * std::pow(span, a.split_x1000 * 1e-3)));
    int stride = last ? 1 : (int)std::max<long long>(1, n_tiles / t_next);
    int skip_n = 0;
    long long pass_tiles = t_next;
    if (last && seeded) { stride = seed_stride; skip_n = seed_n; pass_tiles = n_tiles - skip_n; if (skipped_tiles) *skipped_tiles = skip_n; }
    // (the append counters are zero here: cleared by dense_lower_bound_kernel, then by each non-final selection)
    if (pass_tiles > 0) MF_TRY(launch_pass(1, pass_tiles, stride, skip_n));
    // the sample pass of a two-pass search keeps its sorted list when the last pass can leave its tiles out
    const bool keep = !last && n_pass == 2 && stride >= 2 && a.skip_sample;
    constexpr int kSelWaves = VDB_SEL_WAVES;
    hipLaunchKernelGGL(select_kth_kernel<kSelWaves>, dim3(b), dim3(kSelWaves * 64), 0, st, (const float*)cand_s, (const int*)cand_i,
                       cand_cnt, last ? 0 : 1, cap, kk, thr, (last || keep) ? (last ? a.list_s : a.seed_s) : (float*)nullptr,
                       (last || keep) ? (last ? a.list_i : a.seed_i) : (int*)nullptr, ovf,
                       (last && seeded) ? (const float*)a.seed_s : (const float*)nullptr,
                       (last && seeded) ? (const int*)a.seed_i : (const int*)nullptr);
    MF_TRY(hipGetLastError());
    if (keep) { seeded = true; seed_stride = stride; seed_n = (int)t_next; }
    t_cur = t_next;
  }
  if (a.force_fallback) MF_TRY(hipMemsetAsync(ovf, 1, 1, st));
  // exact list kernel: returns at once unless the overflow flag is set
  hipLaunchKernelGGL(scan_mfma_topk_kernel, dim3(n_qt_fb * fb.n_splits), dim3(kMfThreads), mf_smem_bytes(), st, rows, n_rows, qb16, b,
                     n_qt_fb, fb.n_splits, fb.tiles_per_split, kk, self0, fb.xcd_map, a.ablate, (const int*)ovf, a.part_s, a.part_i);
  return hipGetLastError();
}

}  // namespace vdbk
