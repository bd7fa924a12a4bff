// k_mfma_ws.hip — instantiations of the 256 x 128 warp-specialised pass kernel (scan_mfma4.h).
#include "mfma_pass.h"
#include "scan_mfma4.h"

namespace vdbk {

hipError_t ws_prepare_device() {
  hipError_t e;
  if ((e = hipFuncSetAttribute((const void*)scan_mfma_ws_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_smem_bytes())) != hipSuccess) return e;
  return hipFuncSetAttribute((const void*)scan_mfma_ws_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_smem_bytes());
}

hipError_t ws_launch_pass(const MfPassArgs& p) {
  if (p.mode == 0)
    hipLaunchKernelGGL((scan_mfma_ws_kernel<0>), dim3(p.n_qt * p.n_splits), dim3(kWsThreads), ws_smem_bytes(), p.st, p.rows, p.n_rows, p.qb16, p.b,
                       p.n_qt, p.n_splits, p.tiles_per_split, p.pass_tiles, p.stride, p.self0, p.xcd_map, (const float*)nullptr, p.dense,
                       (int*)nullptr, (float*)nullptr, (int*)nullptr, 0, p.ovf, 0, 0);
  else
    hipLaunchKernelGGL((scan_mfma_ws_kernel<1>), dim3(p.n_qt * p.n_splits), dim3(kWsThreads), ws_smem_bytes(), p.st, p.rows, p.n_rows, p.qb16, p.b,
                       p.n_qt, p.n_splits, p.tiles_per_split, p.pass_tiles, p.stride, p.self0, p.xcd_map, p.thr, p.dense,
                       p.cand_cnt, p.cand_s, p.cand_i, p.cap, p.ovf, p.ablate, p.skip_n);
  return hipGetLastError();
}

}  // namespace vdbk
