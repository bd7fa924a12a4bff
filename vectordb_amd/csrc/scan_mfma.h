// scan_mfma.h — batched (B >= 16) bf16 path on CDNA4 MFMA.  (stub until the kernel lands)
#pragma once
#include <hip/hip_runtime.h>
struct vdb;
namespace vdbk {
inline bool mfma_path_supported(int) { return false; }
}
namespace {
inline int mfma_search(vdb*, const float*, int, int, int, int64_t, int64_t*, float*, hipStream_t) { return -5; }
}
