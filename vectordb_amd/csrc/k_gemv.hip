// k_gemv.hip — instantiations of the HBM-bound scan kernel (scan_gemv.h) and their dispatch.
#include "launch.h"
#include "scan_gemv.h"

namespace vdbk {
namespace {

template <typename T, int DIM, int QB, int E>
void launch_scan_t(const ScanArgs& a, bool nt) {
  if (nt)
    hipLaunchKernelGGL((scan_topk_kernel<T, DIM, QB, E, true>), dim3(a.grid), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i);
  else
    hipLaunchKernelGGL((scan_topk_kernel<T, DIM, QB, E, false>), dim3(a.grid), dim3(kScanThreads), a.smem, a.st, a.rows, a.n_rows, a.qn, a.kk, a.self0, a.allow, a.part_s, a.part_i);
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

bool launch_scan_any(bool bf16, const ScanArgs& a, int ld, int e, bool nt) {
  return bf16 ? launch_scan_generic<bf16_t>(a, ld, e, nt) : launch_scan_generic<float>(a, ld, e, nt);
}

}  // namespace vdbk
