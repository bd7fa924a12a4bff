// mfma_pass.h — host-side seam between k_mfma.hip (the batched search's orchestration, selection and fallback kernels)
// and the pass-kernel families, each of which instantiates in its own translation unit
// (k_mfma_wide16.hip + k_mfma_wide16_diag.hip, k_stream.hip): the library builds in parallel and a change to one kernel rebuilds one file.
#pragma once
#include <hip/hip_runtime.h>

namespace vdbk {

// one pass of a tiled pass kernel over pass_tiles protocol tiles (pass_tile(i, stride, skip_n)); mode 0 = dense sample, 1 = append
struct MfPassArgs {
  int mode; const unsigned char* rows; int n_rows; const unsigned char* qb16; int b;
  int n_qt, n_splits, tiles_per_split, xcd_map;      // grid: n_qt query tiles x n_splits corpus splits
  int pass_tiles, stride, skip_n, self0;
  const float* thr; float* dense; int* cand_cnt; float* cand_s; int* cand_i; int cap;
  int* ovf; int ablate; hipStream_t st;
  int narrow = 0;   // wide16 only: 1 = its 128-wide query tile (n_qt counts tiles of 128 queries then)
};
constexpr int kM5BNC = 256;     // queries per block of the wide tile (scan_mfma5.h asserts it)
constexpr int kStMaxBC = 64;    // queries the streaming kernel serves at most (scan_stream.h asserts it)

hipError_t wide16_diag_prepare_device(); hipError_t wide16_diag_launch_pass(const MfPassArgs& p);   // its ablation builds (k_mfma_wide16_diag.hip)
hipError_t wide16_prepare_device();   hipError_t wide16_launch_pass(const MfPassArgs& p);    // scan_mfma5.h / scan_mfma6.h: 256 x 256, and (narrow) 256 x 128
// scan_stream.h: <= 64 queries, corpus straight from HBM (the grid is the kernel's own: n_qt/n_splits/xcd_map unused)
hipError_t stream_prepare();
hipError_t stream_pass(const MfPassArgs& p, int n_cu, int nt);

}  // namespace vdbk
