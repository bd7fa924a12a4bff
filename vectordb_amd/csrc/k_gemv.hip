// k_gemv.hip — instantiations of the HBM-bound scan kernel (scan_gemv.h) and their dispatch.
#include "launch.h"
#include "scan_gemv.h"
#include "scan_fused.h"

namespace vdbk {
namespace {

template <typename T, int DIM, int QB, int E>
void launch_scan_t(const ScanArgs& a, bool nt) {
  if (nt)
    hipLaunchKernelGGL((scan_topk_kernel<T, DIM, QB, E, true>), dim3(a.grid, a.ny > 0 ? a.ny : 1), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i, a.xcd_slices);
  else
    hipLaunchKernelGGL((scan_topk_kernel<T, DIM, QB, E, false>), dim3(a.grid, a.ny > 0 ? a.ny : 1), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i, a.xcd_slices);
}

template <typename T, int DIM>
bool launch_scan_dim(const ScanArgs& a, int qb, int e, bool nt) {
  if (qb == 1) {
    switch (e) {
      case 1: launch_scan_t<T, DIM, 1, 1>(a, nt); return true;
      case 2: launch_scan_t<T, DIM, 1, 2>(a, nt); return true;
      case 4: launch_scan_t<T, DIM, 1, 4>(a, nt); return true;
      case 8: launch_scan_t<T, DIM, 1, 8>(a, nt); return true;
      case 16: launch_scan_t<T, DIM, 1, 16>(a, nt); return true;
    }
  } else if (qb == 2) {
    switch (e) {
      case 1: launch_scan_t<T, DIM, 2, 1>(a, nt); return true;
      case 2: launch_scan_t<T, DIM, 2, 2>(a, nt); return true;
    }
  } else if (qb == 4) {
    switch (e) {
      case 1: launch_scan_t<T, DIM, 4, 1>(a, nt); return true;
      case 2: launch_scan_t<T, DIM, 4, 2>(a, nt); return true;
    }
  }
  return false;
}

template <typename T>
bool launch_scan_generic(const ScanArgs& a, int ld, int e, bool nt) {
#define GEN_CASE(EE)                                                                                              \
  case EE:                                                                                                        \
    if (nt) hipLaunchKernelGGL((scan_topk_generic_kernel<T, EE, true>), dim3(a.grid), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, ld, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i); \
    else hipLaunchKernelGGL((scan_topk_generic_kernel<T, EE, false>), dim3(a.grid), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, ld, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i);   \
    return true;
  switch (e) { GEN_CASE(1) GEN_CASE(2) GEN_CASE(4) GEN_CASE(8) GEN_CASE(16) }
#undef GEN_CASE
  return false;
}

}  // namespace

bool launch_scan_fast(bool bf16, int dim, const ScanArgs& a, int qb, int e, bool nt) {
  if (dim == 768) return bf16 ? launch_scan_dim<bf16_t, 768>(a, qb, e, nt) : launch_scan_dim<float, 768>(a, qb, e, nt);
  if (dim == 384) return bf16 ? launch_scan_dim<bf16_t, 384>(a, qb, e, nt) : launch_scan_dim<float, 384>(a, qb, e, nt);
  return false;
}

namespace {
template <typename T, int DIM, int QB>
void launch_fused_t(const FusedScanArgs& a, bool nt, int grid, hipStream_t st) {
  const size_t smem = fused_smem_bytes<DIM>(QB, a.kk);
  if (nt) hipLaunchKernelGGL((scan_topk_fused_kernel<T, DIM, QB, true>), dim3(grid), dim3(kScanThreads), smem, st, a);
  else hipLaunchKernelGGL((scan_topk_fused_kernel<T, DIM, QB, false>), dim3(grid), dim3(kScanThreads), smem, st, a);
}
template <typename T, int DIM>
bool launch_fused_dim(const FusedScanArgs& a, int qb, bool nt, int grid, hipStream_t st) {
  switch (qb) {
    case 1: launch_fused_t<T, DIM, 1>(a, nt, grid, st); return true;
    case 2: launch_fused_t<T, DIM, 2>(a, nt, grid, st); return true;
    case 4: launch_fused_t<T, DIM, 4>(a, nt, grid, st); return true;
  }
  return false;
}
}  // namespace

bool launch_scan_fused(bool bf16, int dim, const FusedScanArgs& a, int qb, bool nt, int grid, hipStream_t st) {
  if (a.kk < 1 || a.kk > 64 || grid < 1 || grid > kFusedMaxParts || (a.kks & 3) || a.kks < a.kk) return false;
  if (dim == 768) return bf16 ? launch_fused_dim<bf16_t, 768>(a, qb, nt, grid, st) : launch_fused_dim<float, 768>(a, qb, nt, grid, st);
  if (dim == 384) return bf16 ? launch_fused_dim<bf16_t, 384>(a, qb, nt, grid, st) : launch_fused_dim<float, 384>(a, qb, nt, grid, st);
  return false;
}

bool launch_scan_any(bool bf16, const ScanArgs& a, int ld, int e, bool nt) {
  return bf16 ? launch_scan_generic<bf16_t>(a, ld, e, nt) : launch_scan_generic<float>(a, ld, e, nt);
}

}  // namespace vdbk
