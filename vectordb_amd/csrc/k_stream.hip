// k_stream.hip — instantiations of the streaming pass kernel (scan_stream.h): <= 64 queries, corpus straight from HBM.
#include "mfma_pass.h"
#include "scan_stream.h"

namespace vdbk {

static_assert(kStMaxBC == kStMaxB, "mfma_pass.h mirrors the kernel's batch limit");

hipError_t stream_prepare() { return stream_prepare_device(); }

hipError_t stream_pass(const MfPassArgs& p, int n_cu, int nt) {
  stream_launch_pass(p.mode, p.rows, p.n_rows, p.qb16, p.b, p.pass_tiles, p.stride, p.skip_n, (p.self0 < 0 && (p.ablate & 128)) ? -2 : p.self0, p.mode ? p.thr : nullptr, p.dense,
                     p.cand_cnt, p.cand_s, p.cand_i, p.cap, n_cu, nt, p.st);
  return hipGetLastError();
}

}  // namespace vdbk
