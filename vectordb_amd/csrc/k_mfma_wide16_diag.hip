// k_mfma_wide16_diag.hip — the diagnostic instantiations of the wide pass kernel (scan_mfma6.h): no staging loads, no
// fragment reads / MFMAs, in-kernel cycle stamps (mfma_ablate bits 1, 2, 8; results invalid).  A translation unit of
// their own so that they compile beside the production kernel, not behind it.
#include "mfma_pass.h"
#include "scan_mfma6.h"

namespace vdbk {

hipError_t wide16_diag_prepare_device() {
  hipError_t e;
#define M6_ATTR(...) if ((e = hipFuncSetAttribute((const void*)scan_mfma_wide16_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m5_smem_bytes())) != hipSuccess) return e;
  M6_ATTR(1, 1) M6_ATTR(1, 2) M6_ATTR(1, 3)
#undef M6_ATTR
  return hipSuccess;
}

hipError_t wide16_diag_launch_pass(const MfPassArgs& p) {
  auto kern = (p.ablate & 8) ? scan_mfma_wide16_kernel<1, 3> : (p.ablate & 1) ? scan_mfma_wide16_kernel<1, 1> : scan_mfma_wide16_kernel<1, 2>;
  hipLaunchKernelGGL(kern, dim3(p.n_qt * p.n_splits), dim3(kM5Threads), m5_smem_bytes(), p.st, p.rows, p.n_rows, p.qb16, p.b,
                     p.n_qt, p.n_splits, p.tiles_per_split, p.pass_tiles, p.stride, p.self0, p.xcd_map, p.thr, p.dense,
                     p.cand_cnt, p.cand_s, p.cand_i, p.cap, p.ablate, p.skip_n);
  return hipGetLastError();
}

}  // namespace vdbk
