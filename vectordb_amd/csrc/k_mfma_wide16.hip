// k_mfma_wide16.hip — instantiations of the 256 x 256 pass kernel on v_mfma_f32_16x16x32_bf16 (scan_mfma6.h, the default).
#include "mfma_pass.h"
#include "scan_mfma6.h"

namespace vdbk {

static_assert(kM5BNC == kM5BN, "mfma_pass.h mirrors the kernel's tile width");

hipError_t wide16_prepare_device() {
  hipError_t e;
#define M6_ATTR(...) if ((e = hipFuncSetAttribute((const void*)scan_mfma_wide16_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)m5_smem_bytes())) != hipSuccess) return e;
  M6_ATTR(0) M6_ATTR(1) M6_ATTR(0, 0, 2) M6_ATTR(1, 0, 2)
#undef M6_ATTR
  return wide16_diag_prepare_device();
}

hipError_t wide16_launch_pass(const MfPassArgs& p) {
  const dim3 grid(p.n_qt * p.n_splits), block(kM5Threads);
#define M6_LAUNCH0(...) hipLaunchKernelGGL((scan_mfma_wide16_kernel<__VA_ARGS__>), grid, block, m5_smem_bytes(), p.st, p.rows, p.n_rows, p.qb16, p.b, \
                       p.n_qt, p.n_splits, p.tiles_per_split, p.pass_tiles, p.stride, p.self0, p.xcd_map, (const float*)nullptr, p.dense, \
                       (int*)nullptr, (float*)nullptr, (int*)nullptr, 0, 0, 0)
#define M6_LAUNCH1(...) hipLaunchKernelGGL((scan_mfma_wide16_kernel<__VA_ARGS__>), grid, block, m5_smem_bytes(), p.st, p.rows, p.n_rows, p.qb16, p.b, \
                       p.n_qt, p.n_splits, p.tiles_per_split, p.pass_tiles, p.stride, p.self0, p.xcd_map, p.thr, p.dense, \
                       p.cand_cnt, p.cand_s, p.cand_i, p.cap, p.ablate, p.skip_n)
  if (p.mode == 0) { if (p.narrow) M6_LAUNCH0(0, 0, 2); else M6_LAUNCH0(0); }
  else if (p.narrow) M6_LAUNCH1(1, 0, 2);
  else if (p.ablate & 11) return wide16_diag_launch_pass(p);   // ablation / cycle-stamp builds of the kernel (k_mfma_wide16_diag.hip)
  else M6_LAUNCH1(1, 0);
#undef M6_LAUNCH0
#undef M6_LAUNCH1
  return hipGetLastError();
}

}  // namespace vdbk
