// vdb_api.hip — C ABI (include/vdb.h) of the MI355X-native cosine top-k engine.
// Host-side orchestration only: all arithmetic is in the HIP kernels of scan_gemv.h (batch-1, HBM-bound;
// instantiated in k_gemv.hip), scan_stream.h / scan_mfma{,3,4,5}.h (batched bf16 path on MFMA; k_mfma.hip) and
// select.h (merges, fp32 rerank, shard merge; here).  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <errno.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vdb.h"
#include "launch.h"
#include "select.h"
#include "synth.h"
#include "probes.h"

using namespace vdbk;

namespace {

thread_local std::string g_err;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// ---- scratch of the entry point that allocates nothing (vdb_search_dev_noalloc) ----------------------------
// Device memory beyond the inputs and outputs of a GEMV-path search: the normalised queries and the two
// generations of partial top-k' lists.  A file-scope array, so that the entry point never calls hipMalloc; every
// GPU the library runs on holds one copy (2.3 MiB) until the process ends.  In host code the array's NAME is
// only a stand-in: the device address is taken once per device with hipGetSymbolAddress (vdb_create, with that
// device current) and only that pointer is ever handed to kernels.
constexpr int kNaMaxB = 8;          // queries per call
constexpr int kNaMaxKK = 64;        // k (or k' when reranking)
constexpr int kNaMaxGrid = 512;     // scan blocks = partial lists per query
constexpr int kNaMaxLd = 4096;      // padded row length
constexpr size_t kNaQBytes = (size_t)kNaMaxB * kNaMaxLd * 4;                               // 128 KiB
constexpr size_t kNaPartElems = (size_t)kNaMaxB * kNaMaxGrid * kNaMaxKK;                  // 1 MiB per array
constexpr size_t kNaPart2Elems = (size_t)kNaMaxB * ((kNaMaxGrid + 31) / 32) * kNaMaxKK;   // after one 32-way fold
constexpr size_t kNaListBytes = kNaQBytes + 8 * kNaPartElems + 8 * kNaPart2Elems;
constexpr size_t kNaScratchBytes = kNaListBytes + 512;   // + the fused kernel's ticket and work-pool heads (kFusedTicketBytes; zero at load; its last block resets them)
static_assert(kFusedTicketBytes <= 512, "ticket area");
static __device__ __align__(256) unsigned char scratch[kNaScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};   // per device: the address of ITS copy of `scratch`

// ---- RCCL, loaded on demand (option "shard_exchange_rccl" of a sharded handle): the library does not link against it ----
// The exchange north_star names — an all-gather of the shards' k results followed by the on-device merge — as an
// alternative to the P2P stores, behind the same C ABI, so that the two can be compared on one handle.  The process'
// already-loaded RCCL is used where there is one (PyTorch-ROCm brings its own), else librccl.so.1 from the ROCm tree.
struct RcclApi {
  void* lib = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
bool load_rccl(std::string* why) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return true;
  void* lib = nullptr;
  for (const char* name : {"librccl.so", "librccl.so.1"}) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
  for (const char* name : {"librccl.so.1", "librccl.so"}) if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  if (!lib) { *why = std::string("RCCL is not loadable: ") + (dlerror() ? dlerror() : "librccl.so not found"); return false; }
  RcclApi a;
  a.lib = lib;
  a.CommInitAll = (decltype(a.CommInitAll))dlsym(lib, "ncclCommInitAll");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
  a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
  a.GroupStart = (decltype(a.GroupStart))dlsym(lib, "ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))dlsym(lib, "ncclGroupEnd");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.GroupStart || !a.GroupEnd) { *why = "RCCL: a collective entry point is missing from the loaded library"; return false; }
  g_rccl = a;
  return true;
}
std::string rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : ("ncclResult " + std::to_string(rc)); }

}  // namespace

struct vdb {
  int dim = 0, ld = 0, dtype = 0, keep_f32 = 0, device = 0;
  int64_t n = 0, cap = 0, row_offset = 0;
  float* rows_f32 = nullptr;      // [cap][ld] normalised fp32 (dtype f32, or rerank copy)
  uint16_t* rows_bf16 = nullptr;  // [cap][ld] bf16 (dtype bf16)
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // D2H of self-join blocks while the next block computes (created on first use)
  std::vector<hipEvent_t> ring;   // timing ring: 3 events (start, scan end, finish) per recorded search
  hipEvent_t* ev = nullptr;       // triple of the search in flight
  int64_t n_timed = 0;            // searches recorded since the ring was (re)set
  bool timing_valid = false;
  int timing_every = 1;           // record the three events on every Nth search only (an event record idles the queue ~6 us)
  int64_t n_calls = 0;
  bool ev_on = true;              // this search is a timed one
  double last_bytes = 0.0;
  int last_path = 0;              // kernel family of the last search: 0 = GEMV scan, 4/5/6 = MFMA pass kernels (launch.h)
  int last_passes = 0;            // corpus passes of the last search (GEMV: launches; MFMA: 1)
  int last_fused = 0;             // the last search ran as one-launch fused scans (scan_fused.h)
  int64_t n_ws_allocs = 0;        // workspace (re)allocations so far (stat "workspace_allocs": vdb_warm's effect can be checked)
  int n_cu = 256;
  // options
  int scan_blocks_per_cu = 2;    // 2 x 256 threads per CU: 96 KiB in flight per CU; 4 measured 3-4 % slower at every shard size (bf16)
  int scan_nt_loads = 1;
  int alloc_f32_first = 0;       // 1: allocate the fp32 copy before the bf16 rows (the order before round 3's placement finding)
  int scan_xcd_slices = 1;       // unfused scan: the blocks of one XCD take adjacent tiles (0: block-by-block round-robin)
  int scan_super_rounds = kFusedTilesPerWaveAndRoundC;     // tiles per wave and super-round (0 = off); fused scan: the block groups' tiles interleave in super-rounds (one window over the corpus); 0 = a contiguous range per group
  int scan_group_launch = 1;    // GEMV batches: all whole groups of four queries in one 2-D launch (0 = one launch per group)
  int batch_mfma_min_b = 2;      // smallest batch the MFMA path may serve (the cost model decides above it)
  bool mfma_attr_set = false;    // dynamic-LDS limits of the MFMA kernels raised on this handle's device
  int mfma_cost_model = 1;       // 0: every supported batch >= batch_mfma_min_b takes the MFMA path (tests)
  int mfma_ablate = 0;   // diagnostic: skip parts of the MFMA kernel (wrong results, timing only)
  int stream_xcd_slices = 1;   // the streaming kernel's blocks of one XCD take adjacent units (0: block-by-block round-robin)
  int stream_nt = 1;     // the streaming kernel's append passes: non-temporal corpus loads (whole lines, read once)
  // workspaces (grown on demand)
  DevBuf w_q, w_part_s, w_part_i, w_part2_s, w_part2_i, w_stage, w_out_i, w_out_d, w_idx, w_mfma;
  DevBuf w_q1, w_part_s1, w_part_i1, w_part2_s1, w_part2_i1;   // second set: slot 1 of the two-phase search
  struct Split { bool pending = false; int b = 0, k = 0, kp = 0, grid = 0; hipEvent_t* ev = nullptr; bool ev_on = false; } split[2];
  DevBuf w_cand_s, w_cand_i, w_cand_cnt, w_thr, w_dense, w_list_s, w_list_i;   // append-path state
  DevBuf w_seed_s, w_seed_i;      // the sample pass's sorted list (seeds the final selection)
  DevBuf w_fb_s, w_fb_i;          // partial lists of the MFMA path's overflow fallback (never shared with a pending split scan)
  int mfma_skip_sample = 1;       // 1: the last of two passes leaves out the sample pass's tiles
  int mfma_one_pass_span = 24;    // a store up to this many times the dense sample takes ONE append pass (round 2: up to 204 times: -20 % per 1024-query batch at 200k rows)
  int mfma_dense_tiles = 0;       // tiles of the dense sample (first threshold): 0 = by store size, else 1..64
  void* pin = nullptr; size_t pin_bytes = 0;                                     // pinned host staging for small host-pointer searches
  DevBuf w_mask;                                                                // host-supplied eligibility bitmap (vdb_search_masked)
  int mfma_append_cap = 16384;   // candidate buffer entries per query (also sets how far a pass may grow: cap / 4k')
  int mfma_split_x1000 = 0;      // two append passes: the sample pass covers span^(x/1000) times the dense sample; 0 = by kernel
                                 // family: 500 (tiled kernels: the geometric mean, the sample pass costs flops) or 600 (streaming
                                 // kernel: the sample pass costs no extra bytes — the last pass leaves its tiles out — and every hit
                                 // of the last pass stalls a wave's load ring: 2.57 -> 2.46 ms at 3 queries, 2.52 -> 2.45 at 64, 10M rows;
                                 // 650 is as fast but fills the sample pass's candidate buffers to half their capacity)
  int mfma_variant = 6;          // 4 = 128-wide query tiles only; 5 = + the 256-wide tile (both scan_mfma6.h)
                                 // where it pads the batch <= 1.5x further; 6 = + the streaming kernel (scan_stream.h) for <= 64 queries
  const float* qb16_for_qn = nullptr; int qb16_for_b = 0;   // prep_queries_mfma left the bf16 image of exactly these queries in w_mfma
  int mfma_last_b = 0;           // batch size of the last MFMA search (locates its overflow flag)
  int mfma_last_skipped = 0;     // tiles the last MFMA search's final pass left out (the sample pass's)
  int mfma_force_fallback = 0;   // test hook: route through the exact list kernel
  int scan_balance = 1;          // the fused kernel deals its tiles out by self-tuning per-XCD shares (XCDs stream at different rates)
  int fuse_small = 1;            // <= 4 queries with k' <= 64 on the fast dims: normalise + scan + fold + rerank in ONE launch (scan_fused.h); see fuses()
  DevBuf w_stamps;               // diagnostic: clock stamps of the fused kernel (option "scan_stamps")
  DevBuf w_ticket;               // the fused kernel's arrival counter: zeroed at create, reset by the kernel's last block
  void* na_scratch = nullptr;    // this device's copy of `scratch` (hipGetSymbolAddress at create)
  // ---- sharded handle (vdb_create_sharded): a router over per-device shards; none of the fields above but
  // dim/ld/dtype/keep_f32/device(= shard 0's)/stream/pin/w_stage/w_out_i/w_gather are used on it
  std::vector<vdb*> shards;
  std::vector<hipEvent_t> shard_done;   // per shard: its result is in the gather buffer
  std::vector<hipEvent_t> shard_aux;    // per shard, created on ITS device (an event is recorded on streams of its own device only): the self-join's "queries ready"
  std::vector<char> shard_direct;       // per shard: its finish kernel stores straight into shard 0's memory (same device or P2P)
  std::vector<void*> comms;             // one RCCL communicator per shard (option shard_exchange_rccl; empty = the P2P exchange)
  DevBuf w_xchg;                        // per SHARD: the all-gather's receive buffer, g packed result blocks
  int shard_force_remote = 0;           // test hook: every shard but the first takes the no-peer-access path (query and result peer copies)
  int shard_serial_streams = 0;         // diagnostic: shard 0 on its own stream like the others (the round-2 form)
  hipEvent_t ev_q = nullptr;            // the queries are ready on the caller's stream
  int64_t per = 0;                      // rows per shard (multiple of 256); 0 = not reserved yet
  DevBuf w_gather;                      // on shard 0's device: [g][b][k] idx (int64) then [g][b][k] dist (fp32)
  int64_t issue_ns_sum = 0, issue_calls = 0;   // host time spent inside sharded searches (stat "sharded_issue_ns_avg" reads and resets)
  int64_t issue_phase_ns[8] = {};              // diagnostic (stat "sharded_issue_phase_<i>"): 0 prologue + ev_q record, 1 set-device + ev_q waits,
                                               // 2 query copy + prep, 3 search_core (launches), 4 result copies + shard_done records, 5 main waits, 6 merge launch
  std::string err;
};

namespace {

int fail(vdb_t* h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                                          \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return fail((h), e_ == hipErrorOutOfMemory ? VDB_ERR_MEMORY : VDB_ERR_DEVICE,               \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                             \
  } while (0)

bool is_sharded(const vdb_t* h) { return !h->shards.empty(); }
#define NOT_ON_SHARDED(h, what) \
  do { if (is_sharded(h)) return fail((h), VDB_ERR_STATE, what ": not available on a sharded handle"); } while (0)

int ensure(vdb_t* h, DevBuf& b, size_t bytes) {
  if (bytes <= b.bytes) return VDB_OK;
  // a workspace may still be in use on a caller's stream: drain the device before freeing it
  if (b.p) { HIP_TRY(h, hipDeviceSynchronize()); HIP_TRY(h, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  size_t want = std::max(bytes, (size_t)4096);
  HIP_TRY(h, hipMalloc(&b.p, want));
  b.bytes = want;
  h->n_ws_allocs++;
  return VDB_OK;
}

int set_device(vdb_t* h) {
  HIP_TRY(h, hipSetDevice(h->device));
  return VDB_OK;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

int grow(vdb_t* h, int64_t want_rows) {
  if (want_rows <= h->cap) return VDB_OK;
  want_rows = (want_rows + 255) / 256 * 256;   // the MFMA kernels read whole 256-row tiles (tail rows are masked)
  const size_t ld = (size_t)h->ld;
  // The scanned array first: where a buffer lands in the heap changes the rate a streaming read of it gets by up to 6 %
  // (benchmarks/placement_probe.hip: 19.2 GB read at 7.07 TB/s at the start of a fresh heap, 6.78 behind a 38.4 GB
  // neighbour — which is where a 12.5M-row store's bf16 rows landed while its fp32 copy was allocated first).  The fp32
  // copy is touched by the rerank's k' rows per query only.  (Option alloc_f32_first = 1: the old order, for A/B.)
  auto grow_f32 = [&]() -> int {
    float* p = nullptr;
    HIP_TRY(h, hipMalloc((void**)&p, sizeof(float) * ld * (size_t)want_rows));
    if (h->n) HIP_TRY(h, hipMemcpyAsync(p, h->rows_f32, sizeof(float) * ld * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->rows_f32) HIP_TRY(h, hipFree(h->rows_f32));
    h->rows_f32 = p;
    return VDB_OK;
  };
  auto grow_b16 = [&]() -> int {
    uint16_t* p = nullptr;
    HIP_TRY(h, hipMalloc((void**)&p, sizeof(uint16_t) * ld * (size_t)want_rows));
    // rows beyond n are read (and masked) by the tiled kernels: keep them finite
    HIP_TRY(h, hipMemsetAsync(p, 0, sizeof(uint16_t) * ld * (size_t)want_rows, h->stream));
    if (h->n) HIP_TRY(h, hipMemcpyAsync(p, h->rows_bf16, sizeof(uint16_t) * ld * (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->rows_bf16) HIP_TRY(h, hipFree(h->rows_bf16));
    h->rows_bf16 = p;
    return VDB_OK;
  };
  int rc;
  const bool b16 = h->dtype == VDB_BF16;
  if (h->alloc_f32_first) {
    if (h->keep_f32 && (rc = grow_f32())) return rc;
    if (b16 && (rc = grow_b16())) return rc;
  } else {
    if (b16 && (rc = grow_b16())) return rc;
    if (h->keep_f32 && (rc = grow_f32())) return rc;
  }
  h->cap = want_rows;
  return VDB_OK;
}

int reserve_for_add(vdb_t* h, int64_t n) {
  if (h->n + n > (int64_t)0x7ffffff0) return fail(h, VDB_ERR_ARG, "add: a shard holds at most 2^31-16 rows");
  if (h->n + n <= h->cap) return VDB_OK;
  int64_t want = h->cap + h->cap / 2;
  if (want < h->n + n) want = h->n + n;
  return grow(h, want);
}

// normalise n raw rows (device, leading dim in_ld) into the store at row h->n
int normalize_into_store(vdb_t* h, const float* raw_dev, int in_ld, int64_t n, hipStream_t st) {
  if (n == 0) return VDB_OK;
  const int blocks = (int)std::min<int64_t>((n + 3) / 4, (int64_t)h->n_cu * 8);
  float* of = h->keep_f32 ? h->rows_f32 + (size_t)h->n * h->ld : nullptr;
  uint16_t* ob = h->dtype == VDB_BF16 ? h->rows_bf16 + (size_t)h->n * h->ld : nullptr;
  // rows held in registers, 16-byte accesses: dim a multiple of 4 up to 2048, aligned input (every call site but a
  // caller's odd device pointer); otherwise the scalar kernel
  const int nv = (h->ld / 4 + 63) / 64;
  if ((h->dim & 3) == 0 && (in_ld & 3) == 0 && h->ld <= 2048 && (((uintptr_t)raw_dev) & 15) == 0) {
    switch (nv <= 1 ? 1 : nv <= 2 ? 2 : nv <= 3 ? 3 : nv <= 4 ? 4 : 8) {
      case 1: hipLaunchKernelGGL((normalize_rows_vec_kernel<1>), dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob); break;
      case 2: hipLaunchKernelGGL((normalize_rows_vec_kernel<2>), dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob); break;
      case 3: hipLaunchKernelGGL((normalize_rows_vec_kernel<3>), dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob); break;
      case 4: hipLaunchKernelGGL((normalize_rows_vec_kernel<4>), dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob); break;
      default: hipLaunchKernelGGL((normalize_rows_vec_kernel<8>), dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob); break;
    }
  } else {
    hipLaunchKernelGGL(normalize_rows_kernel, dim3(blocks), dim3(256), 0, st, raw_dev, in_ld, (long long)n, h->dim, h->ld, of, ob);
  }
  HIP_TRY(h, hipGetLastError());
  return VDB_OK;
}

int entries_for(int kk) {
  int e = (kk + 63) / 64;
  int p = 1;
  while (p < e) p <<= 1;
  return p;  // 1,2,4,8,16
}

template <int E>
void launch_merge_t(const float* is, const int* ii, int n_cand, int cpb, int kk, float* os, int* oi, int gx, int b, hipStream_t st) {
  const size_t smem = (size_t)kMergeWaves * kk * 8;
  hipLaunchKernelGGL((merge_lists_kernel<E>), dim3(gx, b), dim3(kMergeThreads), smem, st, is, ii, n_cand, cpb, kk, os, oi);
}
void launch_merge(int e, const float* is, const int* ii, int n_cand, int cpb, int kk, float* os, int* oi, int gx, int b, hipStream_t st) {
  switch (e) {
    case 1: launch_merge_t<1>(is, ii, n_cand, cpb, kk, os, oi, gx, b, st); break;
    case 2: launch_merge_t<2>(is, ii, n_cand, cpb, kk, os, oi, gx, b, st); break;
    case 4: launch_merge_t<4>(is, ii, n_cand, cpb, kk, os, oi, gx, b, st); break;
    case 8: launch_merge_t<8>(is, ii, n_cand, cpb, kk, os, oi, gx, b, st); break;
    default: launch_merge_t<16>(is, ii, n_cand, cpb, kk, os, oi, gx, b, st); break;
  }
}

// Where a GEMV-path search keeps its partial lists: two generations [q][part][kk] of (score, row).  The handle owns
// two such sets (slot 1 lets phase 2 of one batch run on another stream while phase 1 of the next fills slot 0);
// the entry point that allocates nothing brings a third, carved out of the device's `scratch` array.
struct WsPtrs { float* ps; int* pi; float* p2s; int* p2i; };
struct WsSet { DevBuf *q, *ps, *pi, *p2s, *p2i; };
WsSet ws_of(vdb_t* h, int slot) {
  return slot ? WsSet{&h->w_q1, &h->w_part_s1, &h->w_part_i1, &h->w_part2_s1, &h->w_part2_i1}
              : WsSet{&h->w_q, &h->w_part_s, &h->w_part_i, &h->w_part2_s, &h->w_part2_i};
}
WsPtrs ptrs_of(const WsSet& W) { return WsPtrs{(float*)W.ps->p, (int*)W.pi->p, (float*)W.p2s->p, (int*)W.p2i->p}; }

// Reduce the [b][n_parts][kk] partial lists of one query chunk to the b x k outputs:
// hierarchical merge, then fp32 rerank (kp != 0) or plain ranking.
int finish_search(vdb_t* h, const WsPtrs& W, int b, int n_parts, int kk, int k, int kp, const float* qn, int64_t* oi, float* od,
                  hipStream_t st) {
  float* cur_s = W.ps;
  int* cur_i = W.pi;
  float* oth_s = W.p2s;
  int* oth_i = W.p2i;
  const int e = entries_for(kk);
  int lists = n_parts;
  if (e == 1) {
    // sorted one-entry-per-lane lists: bitonic folds, 32 lists per block, then one fused finish launch
    // (letting the finish kernel fold up to 256 lists itself measured slower: 18.9 vs 17.0 us at 231 lists)
    while (lists > 64) {
      const int gx = (lists + 31) / 32;
      hipLaunchKernelGGL(merge_sorted_lists_kernel, dim3(gx, b), dim3(256), 0, st, cur_s, cur_i, lists, 32, kk, oth_s, oth_i);
      HIP_TRY(h, hipGetLastError());
      std::swap(cur_s, oth_s);
      std::swap(cur_i, oth_i);
      lists = gx;
    }
    if (kp)
      hipLaunchKernelGGL((final_sorted_kernel<true>), dim3(b), dim3(1024), 0, st, cur_s, cur_i, lists, kk, k,
                         (long long)h->row_offset, h->keep_f32 ? (const float*)h->rows_f32 : (const float*)nullptr, (const uint16_t*)h->rows_bf16, h->ld, qn, (long long*)oi, od);
    else
      hipLaunchKernelGGL((final_sorted_kernel<false>), dim3(b), dim3(1024), 0, st, cur_s, cur_i, lists, kk, k,
                         (long long)h->row_offset, (const float*)nullptr, (const uint16_t*)nullptr, h->ld, qn, (long long*)oi, od);
    HIP_TRY(h, hipGetLastError());
    return VDB_OK;
  }
  while (lists > 1) {
    const int lists_per_block = std::max(2, std::min(lists, std::max(32, 4096 / kk)));
    const int gx = (lists + lists_per_block - 1) / lists_per_block;
    launch_merge(e, cur_s, cur_i, lists * kk, lists_per_block * kk, kk, oth_s, oth_i, gx, b, st);
    HIP_TRY(h, hipGetLastError());
    std::swap(cur_s, oth_s);
    std::swap(cur_i, oth_i);
    lists = gx;
  }
  if (kp)
    hipLaunchKernelGGL(rerank_kernel, dim3(b), dim3(256), (size_t)kk * 8, st, h->keep_f32 ? (const float*)h->rows_f32 : (const float*)nullptr, (const uint16_t*)h->rows_bf16, h->ld, qn, cur_i, kk, k,
                       (long long)h->row_offset, (long long*)oi, od);
  else
    hipLaunchKernelGGL(finalize_kernel, dim3(b), dim3(256), (size_t)kk * 8, st, cur_s, cur_i, kk, k,
                       (long long)h->row_offset, (long long*)oi, od);
  HIP_TRY(h, hipGetLastError());
  return VDB_OK;
}

// Batched bf16 path (k_mfma.hip): dense sample -> thresholds -> append passes over growing strided samples ->
// full pass -> exact top-k' lists; exact list kernel as the device-flagged overflow fallback; then the shared
// fp32 finish.  At most kMfMaxQueriesPerCall queries per call: the candidate buffers are b x cap x 8 bytes.
int mfma_search(vdb_t* h, const float* qn, int b, int k, int kp, int64_t self_row0, int64_t* out_idx,
                float* out_dist, hipStream_t st, bool last_chunk, bool qb16_ready) {
  const int kk = kp ? kp : k;
  const int n_rows = (int)h->n;
  const MfPlan pl = mfma_plan(b, n_rows, h->n_cu, h->mfma_variant);
  const int cap = h->mfma_append_cap;
  int rc;
  if ((rc = ensure(h, h->w_mfma, (size_t)pl.b_pad * kMfRowBytesC))) return rc;
  if ((rc = ensure(h, h->w_cand_s, (size_t)b * cap * 4))) return rc;
  if ((rc = ensure(h, h->w_cand_i, (size_t)b * cap * 4))) return rc;
  if ((rc = ensure(h, h->w_cand_cnt, ((size_t)b * kMfCntStrideC + 16) * 4))) return rc;   // [b] counters, one cache line each, + the overflow flag
  if ((rc = ensure(h, h->w_thr, (size_t)b * 4))) return rc;
  if ((rc = ensure(h, h->w_dense, (size_t)b * kMfDenseLdC * 4))) return rc;
  if ((rc = ensure(h, h->w_list_s, (size_t)b * kk * 4))) return rc;
  if ((rc = ensure(h, h->w_list_i, (size_t)b * kk * 4))) return rc;
  if ((rc = ensure(h, h->w_seed_s, (size_t)b * kk * 4))) return rc;
  if ((rc = ensure(h, h->w_seed_i, (size_t)b * kk * 4))) return rc;
  const size_t part_elems = (size_t)b * pl.fb_splits * kk;
  if ((rc = ensure(h, h->w_fb_s, part_elems * 4))) return rc;
  if ((rc = ensure(h, h->w_fb_i, part_elems * 4))) return rc;
  if (!h->mfma_attr_set) {   // per handle = per device: the limit is a property of the function ON a device
    HIP_TRY(h, mfma_prepare_device());
    h->mfma_attr_set = true;
  }
  MfSearchArgs a;
  a.rows_bf16 = (const unsigned char*)h->rows_bf16; a.n_rows = n_rows; a.qn = qn; a.b = b; a.kk = kk;
  a.self0 = self_row0 >= 0 ? (int)self_row0 : -1;
  a.n_cu = h->n_cu; a.variant = h->mfma_variant; a.cap = cap; a.split_x1000 = h->mfma_split_x1000 ? h->mfma_split_x1000 : (pl.variant == 6 ? 600 : 500);
  // the first threshold comes from a dense sample of D rows: the two append passes then see about 2 k' sqrt(rows / D) hits
  // per query between them, and hits are what the pass kernels' epilogues pay for — 64 tiles (16 384 rows; 16 in round 2),
  // 128 on large stores where the dense pass is small against the rest (profiles/r03_dense_sample_ab.txt)
  a.dense_tiles = h->mfma_dense_tiles ? h->mfma_dense_tiles : (n_rows >= 6000000 ? 128 : 64);
  a.one_pass_span = h->mfma_one_pass_span;
  a.skip_sample = h->mfma_skip_sample; a.force_fallback = h->mfma_force_fallback; a.ablate = h->mfma_ablate; a.stream_nt = h->stream_nt | (h->stream_xcd_slices << 1);
  a.qb16 = (unsigned char*)h->w_mfma.p;
  a.qb16_ready = qb16_ready;
  a.cand_s = (float*)h->w_cand_s.p; a.cand_i = (int*)h->w_cand_i.p; a.cand_cnt = (int*)h->w_cand_cnt.p;
  a.thr = (float*)h->w_thr.p; a.dense = (float*)h->w_dense.p;
  a.list_s = (float*)h->w_list_s.p; a.list_i = (int*)h->w_list_i.p;
  a.seed_s = (float*)h->w_seed_s.p; a.seed_i = (int*)h->w_seed_i.p;
  a.part_s = (float*)h->w_fb_s.p; a.part_i = (int*)h->w_fb_i.p;
  a.st = st;
  h->mfma_last_b = b;
  h->last_path = pl.variant;
  h->last_passes = 1;
  HIP_TRY(h, mfma_search_launch(a, &h->mfma_last_skipped));
  h->last_bytes = (double)n_rows * kMfRowBytesC;
  if (h->ev_on && last_chunk) HIP_TRY(h, hipEventRecord(h->ev[1], st));
  const int* ovf = a.cand_cnt + (size_t)b * kMfCntStrideC;
  if (kp)
    hipLaunchKernelGGL((final_sorted_kernel<true>), dim3(b), dim3(1024), 0, st, (const float*)a.list_s, (const int*)a.list_i, 1, kk, k,
                       (long long)h->row_offset, h->keep_f32 ? (const float*)h->rows_f32 : (const float*)nullptr, (const uint16_t*)h->rows_bf16, h->ld, qn, (long long*)out_idx, out_dist,
                       ovf, (const float*)a.part_s, (const int*)a.part_i, pl.fb_splits);
  else
    hipLaunchKernelGGL((final_sorted_kernel<false>), dim3(b), dim3(1024), 0, st, (const float*)a.list_s, (const int*)a.list_i, 1, kk, k,
                       (long long)h->row_offset, (const float*)nullptr, (const uint16_t*)nullptr, h->ld, qn, (long long*)out_idx, out_dist,
                       ovf, (const float*)a.part_s, (const int*)a.part_i, pl.fb_splits);
  HIP_TRY(h, hipGetLastError());
  return VDB_OK;
}

// Which path serves a bf16 768-d batch faster?  Coefficients measured on MI355X, refitted in round 3 to the sweeps at 10M and
// 1.25M rows (profiles/r03_batch_sweep.txt: fixed cost + bytes / rate per search, timing events on every call):
//   one query on the fused scan       56 us + bytes / 7.33 TB/s   (2.161 / 0.328 ms)
//   two queries in one fused pass    100 us + bytes / 7.11 TB/s   (2.271 / 0.381 ms: the second query's fold and the fenced
//                                                                  list stores are a fixed cost, not a rate)
//   four queries in one pass          1.64 x a one-query pass (round 2; b = 3 runs as 2 + 1)
//   streaming kernel (<= 64 queries)  80 us + bytes / 7.04 TB/s   (2.272 / 0.362 ms: whole lines through its LDS ring,
//                                                                  dense sample, bound, two selections, finish)
//   tiled kernels                     100 us + 0.25 ns per row and 128-query tile, or 0.36 (+0.05) ns per row and 256-query tile
// So two queries take the streaming kernel below ~8M rows (0.362 vs 0.381 ms at 1.25M) and the scan above, three and
// more the batched path wherever it is allowed.
bool mfma_is_faster(int b, int n_rows, int variant_opt) {
  const double bytes = (double)n_rows * kMfRowBytesC;
  const double pass1 = 56e-6 + bytes / 7.33e12, pass2 = 100e-6 + bytes / 7.11e12;
  const int r = b & 3;
  const double gemv = 1.64 * pass1 * (b >> 2) + (r == 3 ? pass2 + pass1 : r == 2 ? pass2 : r == 1 ? pass1 : 0.0);
  const MfPlan pl = mfma_plan(b, n_rows, 256, variant_opt);
  double mfma;
  if (pl.variant == 6) mfma = 80e-6 + bytes / 7.04e12;
  else if (pl.variant == 5) mfma = 100e-6 + (double)n_rows * (0.36e-9 * pl.n_qt + 0.05e-9);
  else mfma = 100e-6 + (double)n_rows * 0.25e-9 * pl.n_qt;
  return mfma < gemv;
}

constexpr int kQChunk = 64;   // queries per scan/merge round of the GEMV path

bool fast_dim(const vdb_t* h) { return (h->ld == h->dim) && (h->dim == 768 || h->dim == 384); }

// grid: enough blocks to fill the chip, never more tiles than work
int gemv_grid(vdb_t* h) {
  const int rows_per_block_min = 64;
  const int64_t grid = (int64_t)h->n_cu * h->scan_blocks_per_cu;
  return (int)std::max<int64_t>(1, std::min<int64_t>(grid, (h->n + rows_per_block_min - 1) / rows_per_block_min));
}

int ensure_parts(vdb_t* h, const WsSet& W, size_t part_elems) {
  int rc;
  if ((rc = ensure(h, *W.ps, part_elems * 4))) return rc;
  if ((rc = ensure(h, *W.pi, part_elems * 4))) return rc;
  if ((rc = ensure(h, *W.p2s, part_elems * 4))) return rc;
  if ((rc = ensure(h, *W.p2i, part_elems * 4))) return rc;
  return VDB_OK;
}

// The scan passes of one chunk of cb <= 64 normalised queries: partial lists [cb][grid][kk] into W.
int gemv_scan_chunk(vdb_t* h, const WsPtrs& W, const float* qn, int cb, int kk, int64_t self_row0, const unsigned* allow,
                    int grid, hipStream_t st) {
  const int e = entries_for(kk);
  const bool bf16 = h->dtype == VDB_BF16;
  const void* rows = bf16 ? (const void*)h->rows_bf16 : (const void*)h->rows_f32;
  const int n_rows = (int)h->n;
  const bool fast = fast_dim(h);
  int q = 0;
  while (q < cb) {
    int qb = 1, ny = 1;
    if (fast && e <= 2) { if (cb - q >= 4) qb = 4; else if (cb - q >= 2) qb = 2; }
    // all whole groups of four go out as one 2-D launch (a batch against a small store is launch-bound otherwise)
    if (qb == 4 && h->scan_group_launch) ny = (cb - q) / 4;
    ScanArgs a;
    a.rows = rows; a.n_rows = n_rows; a.qn = qn + (size_t)q * h->ld; a.kk = kk;
    a.self0 = self_row0 >= 0 ? (int)(self_row0 + q) : -1;
    a.allow = allow;
    a.part_s = W.ps + (size_t)q * grid * kk;
    a.part_i = W.pi + (size_t)q * grid * kk;
    a.grid = grid; a.st = st; a.ny = ny; a.xcd_slices = h->scan_xcd_slices;
    a.smem = (size_t)4 * qb * kk * 8;   // 4 waves per scan block
    bool ok;
    const bool nt = h->scan_nt_loads != 0;
    if (fast) {
      ok = launch_scan_fast(bf16, h->dim, a, qb, e, nt);
    } else {
      a.smem = (size_t)4 * kk * 8 + (size_t)h->ld * 4;
      ok = launch_scan_any(bf16, a, h->ld, e, nt);
    }
    if (!ok) return fail(h, VDB_ERR_ARG, "search: unsupported k for this path");
    HIP_TRY(h, hipGetLastError());
    h->last_bytes += (double)ny * n_rows * h->ld * (bf16 ? 2 : 4);
    h->last_passes += ny;
    q += qb * ny;
  }
  return VDB_OK;
}

// Does a batch take the MFMA path?  bf16 store WITH an fp32 rerank (the batched kernels round the queries to bf16
// too; the rerank restores fp32 scores, so results do not depend on the batch size — without a rerank every
// batch stays on the GEMV scan, which keeps fp32 queries), fast dim, enough queries that it beats ceil(b/4) GEMV passes.
bool takes_mfma_path(vdb_t* h, int b, int kk, int kp, bool masked) {
  return !masked && kp != 0 && h->dtype == VDB_BF16 && h->dim == 768 && h->ld == 768 && b >= h->batch_mfma_min_b && mfma_kk_supported(kk) &&
         h->n >= 4096 && (!h->mfma_cost_model || mfma_is_faster(b, (int)h->n, h->mfma_variant));
}

// Is this GEMV-path batch served by the fused kernel (scan_fused.h: one launch per <= 4 queries)?
// One query: always (one launch instead of four).  Two to four queries in a pass: the last block folds, re-scores and
// ranks them ONE AFTER THE OTHER (~20 us each) while the rest of the chip idles, which the four-launch form does for all
// queries at once — the fused form wins only where the scan is long (interleaved A/B, profiles/r03_fused_vs_unfused_ab.txt:
// 4 queries 0.104 vs 0.056 ms at 14 777 rows, 0.599 vs 0.572 at 1.25M, 3.60 vs 3.75 at 10M).  fuse_small_batch = 2 forces it.
bool fuses(vdb_t* h, int b, int kk) {
  if (!h->fuse_small || b > 4 || kk > 64 || !fast_dim(h) || gemv_grid(h) > kFusedMaxParts) return false;
  return h->fuse_small >= 2 || b == 1 || (b != 3 && h->n >= 4000000);
}

// The fused launches of b <= 4 queries (4, 2 or 1 per corpus pass): q are raw queries (q_raw: the kernel normalises
// them in its prologue) or unit vectors; each launch writes its queries' k results itself.
int gemv_fused(vdb_t* h, const WsPtrs& W, unsigned* ticket, const float* q, bool q_raw, int b, int kk, int k, int kp,
               int64_t self_row0, const unsigned* allow, int grid, int64_t* out_idx, float* out_dist, hipStream_t st) {
  const bool bf16 = h->dtype == VDB_BF16;
  for (int qd = 0; qd < b;) {
    const int qb = b - qd >= 4 ? 4 : b - qd >= 2 ? 2 : 1;
    FusedScanArgs a;
    a.rows = bf16 ? (const void*)h->rows_bf16 : (const void*)h->rows_f32; a.n_rows = (int)h->n;
    a.q = q + (size_t)qd * h->ld; a.normalize = q_raw ? 1 : 0;
    a.kk = kk; a.k = k; a.kks = round_up(kk, 4);
    a.self0 = self_row0 >= 0 ? (int)(self_row0 + qd) : -1; a.allow = allow;
    a.part_s = W.ps; a.part_i = W.pi;          // the next launch reuses them (stream order)
    a.ticket = ticket; a.balance = h->scan_balance; a.super_rounds = h->scan_super_rounds; a.n_cu = h->n_cu;
    a.rerank = kp != 0; a.rows_f32 = h->keep_f32 ? h->rows_f32 : nullptr; a.rows_b16 = h->rows_bf16;
    a.row_offset = (long long)h->row_offset;
    a.out_idx = (long long*)out_idx + (size_t)qd * k; a.out_dist = out_dist + (size_t)qd * k;
    a.stamps = (long long*)h->w_stamps.p;
    a.probe = 0;
    if (!launch_scan_fused(bf16, h->dim, a, qb, h->scan_nt_loads != 0, grid, st)) return fail(h, VDB_ERR_ARG, "search: unsupported shape for the fused scan");
    HIP_TRY(h, hipGetLastError());
    h->last_bytes += (double)h->n * h->ld * (bf16 ? 2 : 4);
    h->last_passes++;
    qd += qb;
  }
  return VDB_OK;
}

// Core search over device-resident queries qn [b][ld]: already *normalised*, or (q_raw, only where will_fuse()
// says so) raw.
// self_row0 >= 0: query q excludes row self_row0+q.
// allow != nullptr: device bitmap of eligible rows (GEMV scan only).
int search_core(vdb_t* h, const float* qn, int b, int k, int kp, int64_t self_row0, int64_t* out_idx,
                float* out_dist, hipStream_t st, const unsigned* allow = nullptr, bool q_raw = false) {
  const int kk = kp ? kp : k;
  const int n_rows = (int)h->n;
  // prep_queries_mfma's note "the bf16 image of exactly these queries is in place" is good for THIS search only
  const bool qb16_ready = h->qb16_for_qn == qn && h->qb16_for_b == b;
  h->qb16_for_qn = nullptr; h->qb16_for_b = 0;
  // slot 0 of the two-phase search shares these workspaces: a scan pending there would be overwritten
  if (h->split[0].pending || h->split[1].pending)
    return fail(h, VDB_ERR_STATE, "search: a two-phase scan is pending on this handle (call vdb_search_finish_dev first)");

  h->ev_on = h->timing_every <= 1 || (h->n_calls % h->timing_every) == 0;
  h->n_calls++;
  if (h->ev_on) {
    h->ev = h->ring.data() + 3 * (size_t)(h->n_timed % (int64_t)(h->ring.size() / 3));
    h->n_timed++;
    HIP_TRY(h, hipEventRecord(h->ev[0], st));
  }
  h->last_bytes = 0.0;
  h->last_path = 0;
  h->last_passes = 0;
  h->last_fused = 0;

  if (n_rows == 0) {
    // empty store: all -1 / +inf
    hipLaunchKernelGGL(fill_empty_kernel, dim3((b * k + 255) / 256), dim3(256), 0, st, (long long*)out_idx, out_dist, (long long)b * k);
    HIP_TRY(h, hipGetLastError());
    if (h->ev_on) HIP_TRY(h, hipEventRecord(h->ev[1], st));
    if (h->ev_on) HIP_TRY(h, hipEventRecord(h->ev[2], st));
    if (h->ev_on) h->timing_valid = true;
    return VDB_OK;
  }

  int rc;
  if (takes_mfma_path(h, b, kk, kp, allow != nullptr)) {
    // chunks of at most 4096 queries: the candidate buffers are b x cap x 8 bytes (131 KB per query)
    for (int q0 = 0; q0 < b; q0 += kMfMaxQueriesPerCall) {
      const int cb = std::min(kMfMaxQueriesPerCall, b - q0);
      if ((rc = mfma_search(h, qn + (size_t)q0 * h->ld, cb, k, kp, self_row0 >= 0 ? self_row0 + q0 : -1, out_idx + (size_t)q0 * k,
                            out_dist + (size_t)q0 * k, st, q0 + cb == b, qb16_ready && cb == b))) return rc;
    }
    if (h->ev_on) { HIP_TRY(h, hipEventRecord(h->ev[2], st)); h->timing_valid = true; }
    return VDB_OK;
  }

  const int grid = gemv_grid(h);
  const WsSet W = ws_of(h, 0);
  if (fuses(h, b, kk)) {
    if ((rc = ensure_parts(h, W, (size_t)std::min(b, 4) * grid * round_up(kk, 4)))) return rc;
    if ((rc = gemv_fused(h, ptrs_of(W), (unsigned*)h->w_ticket.p, qn, q_raw, b, kk, k, kp, self_row0, allow, grid, out_idx, out_dist, st))) return rc;
    h->last_fused = 1;
    if (h->ev_on) { HIP_TRY(h, hipEventRecord(h->ev[1], st)); HIP_TRY(h, hipEventRecord(h->ev[2], st)); h->timing_valid = true; }
    return VDB_OK;
  }
  if (q_raw) return fail(h, VDB_ERR_STATE, "search: internal: raw queries on an unfused path");
  // queries are processed in chunks so the partial buffers stay small
  const int cb_max = std::min(b, kQChunk);
  if ((rc = ensure_parts(h, W, (size_t)cb_max * grid * kk))) return rc;
  const WsPtrs P = ptrs_of(W);
  for (int q0 = 0; q0 < b; q0 += kQChunk) {
    const int cb = std::min(kQChunk, b - q0);
    if ((rc = gemv_scan_chunk(h, P, qn + (size_t)q0 * h->ld, cb, kk, self_row0 >= 0 ? self_row0 + q0 : -1, allow, grid, st))) return rc;
    if (q0 == 0 && b <= kQChunk) if (h->ev_on) HIP_TRY(h, hipEventRecord(h->ev[1], st));
    // --- merge + finish ----------------------------------------------------
    if ((rc = finish_search(h, P, cb, grid, kk, k, kp, qn + (size_t)q0 * h->ld, out_idx + (size_t)q0 * k,
                            out_dist + (size_t)q0 * k, st))) return rc;
  }
  if (b > kQChunk) if (h->ev_on) HIP_TRY(h, hipEventRecord(h->ev[1], st));  // multi-chunk: scan/merge interleave, report total as scan
  if (h->ev_on) HIP_TRY(h, hipEventRecord(h->ev[2], st));
  if (h->ev_on) h->timing_valid = true;
  return VDB_OK;
}

int resize_ring(vdb_t* h, int cap) {
  for (hipEvent_t e : h->ring) if (e) (void)hipEventDestroy(e);
  h->ring.assign((size_t)cap * 3, nullptr);
  h->ev = nullptr; h->n_timed = 0; h->timing_valid = false;
  for (hipEvent_t& e : h->ring) HIP_TRY(h, hipEventCreate(&e));
  return VDB_OK;
}

int check_search_args(vdb_t* h, const void* q, int b, int k, int kp, const void* oi, const void* od) {
  if (b < 0 || (b > 0 && (!q || !oi || !od))) return fail(h, VDB_ERR_ARG, "search: NULL buffer");
  if (k < 1 || k > VDB_MAX_K) return fail(h, VDB_ERR_ARG, "search: k must be in [1,1024]");
  if (kp != 0 && (kp < k || kp > VDB_MAX_K)) return fail(h, VDB_ERR_ARG, "search: rerank_kprime must be 0 or in [k,1024]");
  if (kp != 0 && h->dtype != VDB_BF16) return fail(h, VDB_ERR_STATE, "search: rerank_kprime needs a bf16 store");   // without an fp32 copy the k' candidates are re-scored against the bf16 rows
  return VDB_OK;
}

// normalise raw device queries [b][dim] into the handle's query workspace [b][ld]
int prep_queries(vdb_t* h, const float* q_dev, int b, hipStream_t st, float** qn, int slot = 0) {
  DevBuf& wq = slot ? h->w_q1 : h->w_q;
  int rc = ensure(h, wq, (size_t)std::max(b, 1) * h->ld * 4);
  if (rc) return rc;
  float* out = (float*)wq.p;
  const int blocks = std::max(1, std::min((b + 3) / 4, h->n_cu * 8));
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(blocks), dim3(256), 0, st, q_dev, h->dim, (long long)b, h->dim, h->ld, out, (uint16_t*)nullptr);
  HIP_TRY(h, hipGetLastError());
  *qn = out;
  return VDB_OK;
}

// The queries of a batch that will take the batched MFMA path as ONE chunk: fp32 normalised copy + bf16 image + cleared
// overflow flag in one launch (mfma_search then skips its own conversion).
int prep_queries_mfma(vdb_t* h, const float* q_dev, int b, hipStream_t st, float** qn) {
  const MfPlan pl = mfma_plan(b, (int)h->n, h->n_cu, h->mfma_variant);
  int rc;
  if ((rc = ensure(h, h->w_q, (size_t)b * h->ld * 4))) return rc;
  if ((rc = ensure(h, h->w_mfma, (size_t)pl.b_pad * kMfRowBytesC))) return rc;
  if ((rc = ensure(h, h->w_cand_cnt, ((size_t)b * kMfCntStrideC + 16) * 4))) return rc;
  hipLaunchKernelGGL(prep_queries_bf16_kernel, dim3(std::max(1, std::min((pl.b_pad + 3) / 4, h->n_cu * 8))), dim3(256), 0, st, q_dev, h->dim, b, pl.b_pad,
                     (float*)h->w_q.p, (uint16_t*)h->w_mfma.p, (int*)h->w_cand_cnt.p + (size_t)b * kMfCntStrideC);
  HIP_TRY(h, hipGetLastError());
  *qn = (float*)h->w_q.p;
  h->qb16_for_qn = *qn; h->qb16_for_b = b;
  return VDB_OK;
}

// The queries search_core gets: normalised into the workspace, or — where the fused kernel will serve the batch and
// normalises in its own prologue — the caller's raw ones as they are.
bool will_fuse(vdb_t* h, int b, int k, int kp, bool masked) {
  const int kk = kp ? kp : k;
  return h->n > 0 && !takes_mfma_path(h, b, kk, kp, masked) && fuses(h, b, kk);
}
int prep_or_pass(vdb_t* h, const float* q_dev, int b, int k, int kp, bool masked, hipStream_t st, const float** q_out, bool* raw) {
  *raw = will_fuse(h, b, k, kp, masked);
  if (*raw) { *q_out = q_dev; return VDB_OK; }
  float* qn = nullptr;
  const bool one_mfma_chunk = b <= kMfMaxQueriesPerCall && h->n > 0 && takes_mfma_path(h, b, kp ? kp : k, kp, masked);
  const int rc = one_mfma_chunk ? prep_queries_mfma(h, q_dev, b, st, &qn) : prep_queries(h, q_dev, b, st, &qn);
  *q_out = qn;
  return rc;
}

}  // namespace

// ===========================================================================
extern "C" {

int vdb_create(vdb_t** out, int dim, int dtype, int keep_f32, int device_id) {
  if (!out) return fail(nullptr, VDB_ERR_ARG, "vdb_create: out is NULL");
  if (dim <= 0 || dim > 32768) return fail(nullptr, VDB_ERR_ARG, "vdb_create: dim must be in [1,32768]");   // the generic scan keeps the query in LDS: 128 KiB + lists
  if (dtype != VDB_F32 && dtype != VDB_BF16) return fail(nullptr, VDB_ERR_ARG, "vdb_create: dtype must be 0 (f32) or 1 (bf16)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, VDB_ERR_DEVICE, std::string("vdb_create: no HIP device available (") + hipGetErrorString(e) + "); this library has no CPU fallback");
  if (device_id < 0 || device_id >= ndev) return fail(nullptr, VDB_ERR_ARG, "vdb_create: device_id out of range");
  vdb_t* h = new (std::nothrow) vdb();
  if (!h) return fail(nullptr, VDB_ERR_MEMORY, "vdb_create: out of host memory");
  h->dim = dim; h->ld = round_up(dim, 8); h->dtype = dtype;
  h->keep_f32 = (dtype == VDB_F32) ? 1 : (keep_f32 != 0);
  h->device = device_id;
  hipDeviceProp_t prop;
  if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&prop, device_id) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(nullptr, VDB_ERR_DEVICE, "vdb_create: device initialisation failed");
  }
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (resize_ring(h, 1) != VDB_OK) { vdb_destroy(h); return fail(nullptr, VDB_ERR_DEVICE, "vdb_create: event creation failed"); }
  if (hipMalloc(&h->w_ticket.p, kFusedTicketBytes) != hipSuccess || hipMemset(h->w_ticket.p, 0, kFusedTicketBytes) != hipSuccess) {
    vdb_destroy(h);
    return fail(nullptr, VDB_ERR_MEMORY, "vdb_create: no device memory for the arrival counter");
  }
  h->w_ticket.bytes = kFusedTicketBytes;
  // this device's copy of the no-allocation entry point's scratch array: its address is taken ONCE per device,
  // with the device current (set above); kernels only ever see this pointer, never the array's name
  if (device_id < kMaxDevices) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device_id]) {
      void* ptr = nullptr;
      const hipError_t se = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
      if (se != hipSuccess || !ptr) {
        vdb_destroy(h);
        return fail(nullptr, VDB_ERR_DEVICE, std::string("vdb_create: hipGetSymbolAddress(scratch): ") + hipGetErrorString(se));
      }
      g_scratch_ptr[device_id] = ptr;
    }
    h->na_scratch = g_scratch_ptr[device_id];
  }
  *out = h;
  return VDB_OK;
}

// Row-sharded store over the GPUs of one node in ONE process: the handle routes add/search to
// one child handle per entry of device_ids (an id may repeat: several shards on one GPU).
int vdb_create_sharded(vdb_t** out, int dim, int dtype, int keep_f32, const int* device_ids, int n_devices) {
  if (!out) return fail(nullptr, VDB_ERR_ARG, "vdb_create_sharded: out is NULL");
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(nullptr, VDB_ERR_ARG, "vdb_create_sharded: need 1..64 device ids");
  vdb_t* p = new (std::nothrow) vdb();
  if (!p) return fail(nullptr, VDB_ERR_MEMORY, "vdb_create_sharded: out of host memory");
  auto bail = [&](int code, const std::string& msg) { vdb_destroy(p); g_err = msg; return code; };
  for (int s = 0; s < n_devices; ++s) {
    vdb_t* c = nullptr;
    const int rc = vdb_create(&c, dim, dtype, keep_f32, device_ids[s]);
    if (rc) { const std::string m = g_err; return bail(rc, m); }
    p->shards.push_back(c);
  }
  vdb_t* c0 = p->shards[0];
  p->dim = c0->dim; p->ld = c0->ld; p->dtype = c0->dtype; p->keep_f32 = c0->keep_f32; p->device = c0->device; p->n_cu = c0->n_cu;
  if (hipSetDevice(p->device) != hipSuccess || hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&p->ev_q, hipEventDisableTiming) != hipSuccess)
    return bail(VDB_ERR_DEVICE, "vdb_create_sharded: stream/event creation failed");
  p->shard_done.assign(n_devices, nullptr);
  p->shard_aux.assign(n_devices, nullptr);
  p->shard_direct.assign(n_devices, 0);
  for (int s = 0; s < n_devices; ++s) {
    vdb_t* c = p->shards[s];
    if (hipSetDevice(c->device) != hipSuccess || hipEventCreateWithFlags(&p->shard_done[s], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->shard_aux[s], hipEventDisableTiming) != hipSuccess)
      return bail(VDB_ERR_DEVICE, "vdb_create_sharded: event creation failed");
    if (c->device == p->device) { p->shard_direct[s] = 1; continue; }
    // P2P: the shard's finish kernel stores its k results straight into shard 0's gather buffer over xGMI
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, c->device, p->device) == hipSuccess && can) {
      const hipError_t pe = hipDeviceEnablePeerAccess(p->device, 0);
      if (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) p->shard_direct[s] = 1;
    }
    (void)hipGetLastError();
  }
  *out = p;
  return VDB_OK;
}

void vdb_destroy(vdb_t* h) {
  if (!h) return;
  for (vdb_t* c : h->shards) vdb_destroy(c);
  for (size_t s = 0; s < h->shard_done.size(); ++s) if (h->shard_done[s]) (void)hipEventDestroy(h->shard_done[s]);
  for (size_t s = 0; s < h->shard_aux.size(); ++s) if (h->shard_aux[s]) (void)hipEventDestroy(h->shard_aux[s]);
  for (void* c : h->comms) if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
  if (h->ev_q) (void)hipEventDestroy(h->ev_q);
  DevBuf* bufs[] = {&h->w_xchg, &h->w_stamps, &h->w_ticket, &h->w_gather, &h->w_fb_s, &h->w_fb_i, &h->w_q1, &h->w_part_s1, &h->w_part_i1, &h->w_part2_s1, &h->w_part2_i1, &h->w_q, &h->w_part_s, &h->w_part_i, &h->w_part2_s, &h->w_part2_i, &h->w_stage, &h->w_out_i, &h->w_out_d, &h->w_idx, &h->w_mfma,
                    &h->w_cand_s, &h->w_cand_i, &h->w_cand_cnt, &h->w_thr, &h->w_dense, &h->w_list_s, &h->w_list_i, &h->w_mask, &h->w_seed_s, &h->w_seed_i};
  for (DevBuf* b : bufs) if (b->p) (void)hipFree(b->p);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->rows_f32) (void)hipFree(h->rows_f32);
  if (h->rows_bf16) (void)hipFree(h->rows_bf16);
  for (hipEvent_t e : h->ring) if (e) (void)hipEventDestroy(e);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* vdb_last_error(vdb_t* h) { return h ? h->err.c_str() : g_err.c_str(); }

int vdb_reserve(vdb_t* h, int64_t n_rows) {
  if (!h) return VDB_ERR_ARG;
  if (is_sharded(h)) {
    // fixes the partition: shard s owns global rows [s*per, (s+1)*per), the last shard also everything beyond
    const int64_t g = (int64_t)h->shards.size();
    if (n_rows < 0 || n_rows > g * (int64_t)0x7fffff00) return fail(h, VDB_ERR_ARG, "reserve: n_rows out of range");
    const int64_t per = std::max<int64_t>(256, ((n_rows + g - 1) / g + 255) / 256 * 256);
    if (h->per && h->n > 0 && per != h->per) {
      if (n_rows <= h->per * g) return VDB_OK;
      return fail(h, VDB_ERR_STATE, "reserve: a sharded handle's partition is fixed by the first vdb_reserve once rows are stored");
    }
    h->per = per;
    for (int64_t s = 0; s < g; ++s) {
      vdb_t* c = h->shards[(size_t)s];
      c->row_offset = s * per;
      const int64_t want = std::max<int64_t>(0, std::min<int64_t>(per, n_rows - s * per));
      const int rc = vdb_reserve(c, want);
      if (rc) return fail(h, rc, c->err);
    }
    return VDB_OK;
  }
  if (n_rows < 0 || n_rows > (int64_t)0x7ffffff0) return fail(h, VDB_ERR_ARG, "reserve: n_rows out of range");
  int rc = set_device(h);
  if (rc) return rc;
  return grow(h, n_rows);
}

// Re-partition a sharded handle for total_rows rows: the stored rows move between the shards' GPUs as they are (device
// to device, verbatim bits), so that a store that outgrew the partition vdb_reserve fixed — its last shard takes
// every row beyond it — is balanced again.  Searches return what they returned before.
int vdb_rebalance(vdb_t* h, int64_t total_rows) {
  if (!h) return VDB_ERR_ARG;
  if (total_rows < 0) return fail(h, VDB_ERR_ARG, "rebalance: total_rows must be >= 0");
  if (!is_sharded(h)) return vdb_reserve(h, std::max(total_rows, h->n));   // one shard: capacity is all there is to fix
  const int64_t g = (int64_t)h->shards.size();
  const int64_t total = std::max(total_rows, h->n);
  if (total > g * (int64_t)0x7fffff00) return fail(h, VDB_ERR_ARG, "rebalance: total_rows out of range");
  const int64_t per = std::max<int64_t>(256, ((total + g - 1) / g + 255) / 256 * 256);
  if (h->n == 0) h->per = 0;                                 // nothing stored: a fresh partition
  if (h->n == 0 || per == h->per) return vdb_reserve(h, total);
  for (vdb_t* c : h->shards)
    if (c->split[0].pending || c->split[1].pending) return fail(h, VDB_ERR_STATE, "rebalance: a two-phase scan is pending");
  int rc;
  for (vdb_t* c : h->shards) { if ((rc = set_device(c))) return fail(h, rc, c->err); HIP_TRY(h, hipDeviceSynchronize()); }
  const int64_t old_per = h->per;
  struct NewRows { float* f32 = nullptr; uint16_t* b16 = nullptr; int64_t n = 0, cap = 0; };
  std::vector<NewRows> nr((size_t)g);
  auto drop = [&]() { for (int64_t s = 0; s < g; ++s) { (void)hipSetDevice(h->shards[(size_t)s]->device); if (nr[(size_t)s].f32) (void)hipFree(nr[(size_t)s].f32); if (nr[(size_t)s].b16) (void)hipFree(nr[(size_t)s].b16); } };
  const size_t ld = (size_t)h->ld;
  for (int64_t s = 0; s < g; ++s) {
    vdb_t* c = h->shards[(size_t)s];
    const int64_t lo = s * per, hi = s == g - 1 ? std::max(h->n, lo) : std::min(h->n, (s + 1) * per);
    NewRows& w = nr[(size_t)s];
    w.n = std::max<int64_t>(0, hi - lo);
    w.cap = (std::max(per, w.n) + 255) / 256 * 256;
    if (set_device(c) != VDB_OK) { drop(); return fail(h, VDB_ERR_DEVICE, c->err); }
    hipError_t e = hipSuccess;
    if (c->dtype == VDB_BF16) {   // the scanned array first, as in grow()
      e = hipMalloc((void**)&w.b16, sizeof(uint16_t) * ld * (size_t)w.cap);
      if (e == hipSuccess) e = hipMemset(w.b16, 0, sizeof(uint16_t) * ld * (size_t)w.cap);   // rows beyond n are read (and masked) by the tiled kernels
    }
    if (e == hipSuccess && c->keep_f32) e = hipMalloc((void**)&w.f32, sizeof(float) * ld * (size_t)w.cap);
    // the old shards whose rows [o * old_per, o * old_per + n_o) overlap [lo, lo + w.n)
    for (int64_t o = 0; e == hipSuccess && o < g && w.n > 0; ++o) {
      vdb_t* src = h->shards[(size_t)o];
      const int64_t olo = o * old_per, a = std::max(lo, olo), b = std::min(lo + w.n, olo + src->n);
      if (b <= a) continue;
      if (w.f32) e = hipMemcpyPeer(w.f32 + (size_t)(a - lo) * ld, c->device, src->rows_f32 + (size_t)(a - olo) * ld, src->device, sizeof(float) * ld * (size_t)(b - a));
      if (e == hipSuccess && w.b16) e = hipMemcpyPeer(w.b16 + (size_t)(a - lo) * ld, c->device, src->rows_bf16 + (size_t)(a - olo) * ld, src->device, sizeof(uint16_t) * ld * (size_t)(b - a));
    }
    if (e != hipSuccess) { drop(); return fail(h, e == hipErrorOutOfMemory ? VDB_ERR_MEMORY : VDB_ERR_DEVICE, std::string("rebalance: ") + hipGetErrorString(e)); }
  }
  for (int64_t s = 0; s < g; ++s) {
    vdb_t* c = h->shards[(size_t)s];
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->rows_f32) (void)hipFree(c->rows_f32);
    if (c->rows_bf16) (void)hipFree(c->rows_bf16);
    c->rows_f32 = nr[(size_t)s].f32; c->rows_bf16 = nr[(size_t)s].b16;
    c->n = nr[(size_t)s].n; c->cap = nr[(size_t)s].cap; c->row_offset = s * per;
  }
  h->per = per;
  return set_device(h);
}

namespace {
// The shard that owns global row r of a sharded handle, and the rows [r, r+len) of an append that land in it.
struct ShardSpan { int s; int64_t count; };
ShardSpan shard_span(const vdb_t* h, int64_t r, int64_t len) {
  const int g = (int)h->shards.size();
  const int s = (int)std::min<int64_t>(r / h->per, g - 1);
  const int64_t end = s == g - 1 ? r + len : std::min<int64_t>(r + len, (int64_t)(s + 1) * h->per);
  return ShardSpan{s, end - r};
}
}  // namespace

int vdb_set_row_offset(vdb_t* h, int64_t off) {
  if (!h) return VDB_ERR_ARG;
  NOT_ON_SHARDED(h, "vdb_set_row_offset");
  h->row_offset = off;
  return VDB_OK;
}

int vdb_add(vdb_t* h, const float* rows_host, int64_t n) {
  if (!h) return VDB_ERR_ARG;
  if (n < 0 || (n > 0 && !rows_host)) return fail(h, VDB_ERR_ARG, "vdb_add: bad rows/n");
  if (n == 0) return VDB_OK;
  if (is_sharded(h)) {
    if (!h->per) return fail(h, VDB_ERR_STATE, "vdb_add: call vdb_reserve(total rows) on a sharded handle first (it fixes the row partition)");
    for (int64_t r = 0; r < n;) {
      const ShardSpan sp = shard_span(h, h->n, n - r);
      vdb_t* c = h->shards[(size_t)sp.s];
      const int rc = vdb_add(c, rows_host + (size_t)r * h->dim, sp.count);
      if (rc) return fail(h, rc, c->err);
      r += sp.count; h->n += sp.count;
    }
    return VDB_OK;
  }
  int rc = set_device(h);
  if (rc) return rc;
  if ((rc = reserve_for_add(h, n))) return rc;
  // stage in bounded chunks: H2D -> normalise(+convert) into the store
  const int64_t chunk_rows = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)h->dim * 4));
  if ((rc = ensure(h, h->w_stage, (size_t)std::min(n, chunk_rows) * h->dim * 4))) return rc;
  for (int64_t r = 0; r < n; r += chunk_rows) {
    const int64_t m = std::min(chunk_rows, n - r);
    HIP_TRY(h, hipMemcpyAsync(h->w_stage.p, rows_host + (size_t)r * h->dim, (size_t)m * h->dim * 4, hipMemcpyHostToDevice, h->stream));
    if ((rc = normalize_into_store(h, (const float*)h->w_stage.p, h->dim, m, h->stream))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // staging buffer is reused
    h->n += m;
  }
  return VDB_OK;
}

int vdb_add_dev(vdb_t* h, const float* rows_dev, int64_t n) {
  if (!h) return VDB_ERR_ARG;
  if (n < 0 || (n > 0 && !rows_dev)) return fail(h, VDB_ERR_ARG, "vdb_add_dev: bad rows/n");
  if (n == 0) return VDB_OK;
  if (is_sharded(h)) {   // the rows live on shard 0's device: each span goes to its shard in place, or over a peer copy
    if (!h->per) return fail(h, VDB_ERR_STATE, "vdb_add_dev: call vdb_reserve(total rows) on a sharded handle first (it fixes the row partition)");
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)h->dim * 4));
    for (int64_t r = 0; r < n;) {
      const ShardSpan sp = shard_span(h, h->n, n - r);
      vdb_t* c = h->shards[(size_t)sp.s];
      const float* src = rows_dev + (size_t)r * h->dim;
      int rc;
      if (c->device == h->device && !(h->shard_force_remote && sp.s > 0)) {
        if ((rc = vdb_add_dev(c, src, sp.count))) return fail(h, rc, c->err);
      } else {
        if ((rc = set_device(c))) return fail(h, rc, c->err);
        if ((rc = ensure(c, c->w_stage, (size_t)std::min(sp.count, chunk) * h->dim * 4))) return fail(h, rc, c->err);
        for (int64_t o = 0; o < sp.count; o += chunk) {
          const int64_t m = std::min(chunk, sp.count - o);
          HIP_TRY(h, hipMemcpyPeer(c->w_stage.p, c->device, src + (size_t)o * h->dim, h->device, (size_t)m * h->dim * 4));
          if ((rc = vdb_add_dev(c, (const float*)c->w_stage.p, m))) return fail(h, rc, c->err);
        }
      }
      r += sp.count; h->n += sp.count;
    }
    return VDB_OK;
  }
  int rc = set_device(h);
  if (rc) return rc;
  if ((rc = reserve_for_add(h, n))) return rc;
  if ((rc = normalize_into_store(h, rows_dev, h->dim, n, h->stream))) return rc;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->n += n;
  return VDB_OK;
}

int vdb_count(vdb_t* h, int64_t* n) {
  if (!h || !n) return VDB_ERR_ARG;
  *n = h->n;
  return VDB_OK;
}

namespace {
// What a sharded search may carry beside the plain arguments.
struct ShardedExtra {
  const uint32_t* allow_host = nullptr;   // where= bitmap over GLOBAL rows, in host memory ...
  const uint32_t* allow_dev0 = nullptr;   // ... or on shard 0's device (one of the two, or neither)
  int64_t self_global0 = -1;              // >= 0: query q must not return global row self_global0 + q (self-join; the block lies in ONE shard)
  bool q_unit_ld = false;                 // the queries are unit vectors laid out [b][ld] (stored rows): used as they are
};

// Search on a sharded handle: queries and outputs live on shard 0's device, `st` is a stream of that device.
// Every shard scans its rows on its own stream (all devices run together), its finish kernel stores the k results
// straight into shard 0's gather buffer (same device, or over xGMI with peer access; a peer copy otherwise), and one
// merge kernel on `st` ranks the g lists.  One host thread, no host synchronisation.  Shard 0 (it lives on the
// caller's device) runs on the CALLER's stream: no cross-stream wait before it and none behind it — a
// hipStreamWaitEvent on a pending event costs the host ~5 us (profiles/r03_sharded_issue_cost.json), and two per
// shard were 60 % of the call.
int sharded_search_dev(vdb_t* h, const float* q_dev0, int b, int k, int kp, int64_t* out_idx, float* out_dist, hipStream_t st,
                       const ShardedExtra& x = ShardedExtra()) {
  const int g = (int)h->shards.size();
  int rc;
  struct IssueClock {   // host time of the whole call, whatever way it returns
    vdb_t* h; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~IssueClock() { h->issue_ns_sum += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); h->issue_calls++; }
  } issue_clock{h};
  if ((rc = set_device(h))) return rc;
  const size_t bk = (size_t)b * k;
  if ((rc = ensure(h, h->w_gather, (size_t)g * bk * 12))) return rc;
  int64_t* g_idx = (int64_t*)h->w_gather.p;
  float* g_dist = (float*)((char*)h->w_gather.p + (size_t)g * bk * 8);
  auto lap = [&](int ph, std::chrono::steady_clock::time_point& t) {
    const auto t1 = std::chrono::steady_clock::now();
    h->issue_phase_ns[ph] += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t).count();
    t = t1;
  };
  auto tp = issue_clock.t0;
  const bool masked = x.allow_host || x.allow_dev0;
  const bool rccl = !h->comms.empty();
  const size_t xwords = bk + (bk + 1) / 2;   // 8-byte words of one shard's packed result block (the all-gather's unit)
  const size_t q_bytes = (size_t)b * (x.q_unit_ld ? h->ld : h->dim) * 4;
  HIP_TRY(h, hipEventRecord(h->ev_q, st));   // the queries are ready; the previous call's merge (earlier on st) has read the gather buffer
  lap(0, tp);
  for (int s = 0; s < g; ++s) {
    vdb_t* c = h->shards[(size_t)s];
    // shard 0 on the caller's stream (stream order does what the two events do for the others); the test hook
    // shard_force_remote treats every other shard as if it lived on another GPU without peer access
    const bool inline0 = s == 0 && !h->shard_serial_streams;
    const bool remote = c->device != h->device || (h->shard_force_remote && s > 0);
    hipStream_t cs = inline0 ? st : c->stream;
    if ((rc = set_device(c))) return fail(h, rc, c->err);
    if (!inline0) HIP_TRY(h, hipStreamWaitEvent(cs, h->ev_q, 0));
    lap(1, tp);
    const float* q_local = q_dev0;
    if (remote) {
      if ((rc = ensure(c, c->w_stage, q_bytes))) return fail(h, rc, c->err);
      HIP_TRY(h, hipMemcpyPeerAsync(c->w_stage.p, c->device, q_dev0, h->device, q_bytes, cs));
      q_local = (const float*)c->w_stage.p;
    }
    // this shard's slice of the eligibility bitmap: its first global row is a multiple of 256, so the slice starts on a word
    const unsigned* allow = nullptr;
    if (masked && c->n > 0) {
      const size_t word0 = (size_t)(c->row_offset >> 5), words = (size_t)((c->n + 31) / 32);
      if (x.allow_dev0 && !remote) {
        allow = x.allow_dev0 + word0;
      } else {
        if ((rc = ensure(c, c->w_mask, words * 4))) return fail(h, rc, c->err);
        if (x.allow_host) HIP_TRY(h, hipMemcpyAsync(c->w_mask.p, x.allow_host + word0, words * 4, hipMemcpyHostToDevice, cs));
        else HIP_TRY(h, hipMemcpyPeerAsync(c->w_mask.p, c->device, x.allow_dev0 + word0, h->device, words * 4, cs));
        allow = (const unsigned*)c->w_mask.p;
      }
    }
    const float* qn = q_local;
    bool raw = false;
    if (!x.q_unit_ld && (rc = prep_or_pass(c, q_local, b, k, kp, masked, cs, &qn, &raw))) return fail(h, rc, c->err);
    int64_t self0 = -1;
    if (x.self_global0 >= 0 && x.self_global0 >= c->row_offset && x.self_global0 < c->row_offset + c->n) self0 = x.self_global0 - c->row_offset;
    int64_t* oi = g_idx + (size_t)s * bk;
    float* od = g_dist + (size_t)s * bk;
    const bool direct = !rccl && h->shard_direct[(size_t)s] && !(h->shard_force_remote && s > 0);
    if (!direct) {
      if ((rc = ensure(c, c->w_out_i, xwords * 8))) return fail(h, rc, c->err);   // packed [idx int64 x bk | dist fp32 x bk | pad to 8]
      oi = (int64_t*)c->w_out_i.p;
      od = (float*)((char*)c->w_out_i.p + bk * 8);
      if (rccl && (rc = ensure(c, c->w_xchg, (size_t)g * xwords * 8))) return fail(h, rc, c->err);
    }
    lap(2, tp);
    if ((rc = search_core(c, qn, b, k, kp, self0, oi, od, cs, allow, raw))) return fail(h, rc, c->err);
    lap(3, tp);
    if (!direct && !rccl) {
      HIP_TRY(h, hipMemcpyPeerAsync(g_idx + (size_t)s * bk, h->device, oi, c->device, bk * 8, cs));
      HIP_TRY(h, hipMemcpyPeerAsync(g_dist + (size_t)s * bk, h->device, od, c->device, bk * 4, cs));
    }
    if (!inline0 && !rccl) HIP_TRY(h, hipEventRecord(h->shard_done[(size_t)s], cs));
    lap(4, tp);
  }
  if (rccl) {
    // ONE all-gather of the packed per-shard results, every shard on its own stream (shard 0 on the caller's): the
    // collective is the only cross-shard ordering — no completion events, no waits on the caller's stream
    int nrc = g_rccl.GroupStart();
    for (int s = 0; s < g && nrc == 0; ++s) {
      vdb_t* c = h->shards[(size_t)s];
      nrc = g_rccl.AllGather(c->w_out_i.p, c->w_xchg.p, xwords * 8, /*ncclChar*/ 0, h->comms[(size_t)s], (s == 0 && !h->shard_serial_streams) ? st : c->stream);
    }
    const int erc = g_rccl.GroupEnd();
    if (nrc == 0) nrc = erc;
    if (nrc != 0) return fail(h, VDB_ERR_DEVICE, "ncclAllGather: " + rccl_err(nrc));
    if ((rc = set_device(h))) return rc;
    vdb_t* c0 = h->shards[0];
    if (h->shard_serial_streams) {   // shard 0 on a stream of its own: the caller's stream waits for its all-gather
      HIP_TRY(h, hipEventRecord(h->shard_done[0], c0->stream));
      HIP_TRY(h, hipStreamWaitEvent(st, h->shard_done[0], 0));
    }
    lap(5, tp);
    hipLaunchKernelGGL(merge_shards_kernel, dim3(b), dim3(256), 0, st, (const float*)((const char*)c0->w_xchg.p + bk * 8), (long long)(2 * xwords),
                       (const long long*)c0->w_xchg.p, (long long)xwords, g, k, (long long*)out_idx, out_dist);
    HIP_TRY(h, hipGetLastError());
    lap(6, tp);
    return VDB_OK;
  }
  if ((rc = set_device(h))) return rc;
  for (int s = h->shard_serial_streams ? 0 : 1; s < g; ++s) HIP_TRY(h, hipStreamWaitEvent(st, h->shard_done[(size_t)s], 0));
  lap(5, tp);
  hipLaunchKernelGGL(merge_shards_kernel, dim3(b), dim3(256), 0, st, (const float*)g_dist, (long long)bk, (const long long*)g_idx,
                     (long long)bk, g, k, (long long*)out_idx, out_dist);
  HIP_TRY(h, hipGetLastError());
  lap(6, tp);
  return VDB_OK;
}
}  // namespace

int vdb_search_dev(vdb_t* h, const float* queries_dev, int b, int k, int kp, int64_t* out_idx_dev,
                   float* out_dist_dev, void* stream) {
  if (!h) return VDB_ERR_ARG;
  int rc = check_search_args(h, queries_dev, b, k, kp, out_idx_dev, out_dist_dev);
  if (rc) return rc;
  if (b == 0) return VDB_OK;
  if (is_sharded(h)) return sharded_search_dev(h, queries_dev, b, k, kp, out_idx_dev, out_dist_dev, (hipStream_t)stream);
  if ((rc = set_device(h))) return rc;
  hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream, HIP convention
  const float* qn = nullptr;
  bool raw = false;
  if ((rc = prep_or_pass(h, queries_dev, b, k, kp, false, st, &qn, &raw))) return rc;
  return search_core(h, qn, b, k, kp, -1, out_idx_dev, out_dist_dev, st, nullptr, raw);
}

// Size every workspace a search of up to max_batch queries with these k / k' needs on the store as it is now, so that
// no later search of that shape grows one (a growth drains the device, frees and allocates): one dummy search per
// kernel family the batch sizes up to max_batch can take, on zero queries, results discarded.
int vdb_warm(vdb_t* h, int max_batch, int k, int kp) {
  if (!h) return VDB_ERR_ARG;
  if (max_batch < 1 || max_batch > (1 << 20)) return fail(h, VDB_ERR_ARG, "warm: max_batch must be in [1, 1048576]");
  int rc = set_device(h);
  if (rc) return rc;
  float* q = nullptr; int64_t* oi = nullptr; float* od = nullptr;
  hipError_t e = hipMalloc((void**)&q, (size_t)max_batch * h->dim * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&oi, (size_t)max_batch * k * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&od, (size_t)max_batch * k * 4);
  rc = e == hipSuccess ? VDB_OK : fail(h, VDB_ERR_MEMORY, std::string("warm: ") + hipGetErrorString(e));
  if (rc == VDB_OK) {   // isotropic random queries (all-zero ones would tie every score and overflow the batched path's candidate buffers)
    vdb_synth_spec sp;
    memset(&sp, 0, sizeof sp);
    sp.seed = 0x5eed; sp.kind = 0;
    const int blocks = (int)std::min<int64_t>(((int64_t)max_batch + 3) / 4, (int64_t)h->n_cu * 8);
    hipLaunchKernelGGL(synth_rows_kernel, dim3(blocks), dim3(256), 0, h->stream, q, 0LL, (long long)max_batch, h->dim, h->dim, sp);
    if (hipGetLastError() != hipSuccess) rc = fail(h, VDB_ERR_DEVICE, "warm: query generation failed");
  }
  // one batch per family boundary: 1..4 (fused scan), 64 (streaming kernel / chunked scan), 128, 256-multiples, the whole batch
  const int sizes[] = {1, 4, 5, 64, 65, 128, 129, 512, max_batch};
  int done_b = 0;
  for (int b : sizes) {
    b = std::min(b, max_batch);
    if (rc != VDB_OK || b <= done_b) continue;
    rc = vdb_search_dev(h, q, b, k, kp, oi, od, h->stream);
    done_b = b;
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess && rc == VDB_OK) rc = fail(h, VDB_ERR_DEVICE, "warm: the dummy searches failed");
  if (is_sharded(h)) for (vdb_t* c : h->shards) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  (void)hipSetDevice(h->device);
  if (q) (void)hipFree(q);
  if (oi) (void)hipFree(oi);
  if (od) (void)hipFree(od);
  return rc;
}

// vdb_search_dev for callers that cannot allocate (HIP graph capture, tight serving loops): the GEMV scan with
// its workspaces carved out of this device's `scratch` array.  No hipMalloc / hipFree / event creation here.
int vdb_search_dev_noalloc(vdb_t* h, const float* queries_dev, int b, int k, int kp, int64_t* out_idx_dev,
                           float* out_dist_dev, void* stream) {
  if (!h) return VDB_ERR_ARG;
  NOT_ON_SHARDED(h, "vdb_search_dev_noalloc");
  int rc = check_search_args(h, queries_dev, b, k, kp, out_idx_dev, out_dist_dev);
  if (rc) return rc;
  if (b == 0) return VDB_OK;
  const int kk = kp ? kp : k;
  if (b > kNaMaxB) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: at most 8 queries per call (the fixed scratch array holds no more)");
  if (kk > kNaMaxKK) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: k (k' when reranking) must be <= 64");
  if (h->ld > kNaMaxLd) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: dim must be <= 4096");
  if (!h->na_scratch) return fail(h, VDB_ERR_STATE, "search_dev_noalloc: no scratch array on this device");
  if (h->split[0].pending || h->split[1].pending) return fail(h, VDB_ERR_STATE, "search_dev_noalloc: a two-phase scan is pending on this handle");
  if ((rc = set_device(h))) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (h->n == 0) {
    hipLaunchKernelGGL(fill_empty_kernel, dim3((b * k + 255) / 256), dim3(256), 0, st, (long long*)out_idx_dev, out_dist_dev, (long long)b * k);
    HIP_TRY(h, hipGetLastError());
    return VDB_OK;
  }
  const int grid = std::min(gemv_grid(h), kNaMaxGrid);
  // carve the device's array (its address, taken at vdb_create — never the array's name): queries | lists | folded lists
  unsigned char* base = (unsigned char*)h->na_scratch;
  float* qn = (float*)base;
  WsPtrs P;
  P.ps = (float*)(base + kNaQBytes);
  P.pi = (int*)(base + kNaQBytes + 4 * kNaPartElems);
  P.p2s = (float*)(base + kNaQBytes + 8 * kNaPartElems);
  P.p2i = (int*)(base + kNaQBytes + 8 * kNaPartElems + 4 * kNaPart2Elems);
  h->ev_on = false;               // no timing events on this path
  h->last_bytes = 0.0; h->last_path = 0; h->last_passes = 0;
  if (fuses(h, b, kk))            // one launch per <= 4 queries, its arrival counter in the same array
    return gemv_fused(h, P, (unsigned*)(base + kNaListBytes), queries_dev, true, b, kk, k, kp, -1, nullptr, grid, out_idx_dev, out_dist_dev, st);
  const int blocks = std::max(1, std::min((b + 3) / 4, h->n_cu * 8));
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(blocks), dim3(256), 0, st, queries_dev, h->dim, (long long)b, h->dim, h->ld, qn, (uint16_t*)nullptr);
  HIP_TRY(h, hipGetLastError());
  if ((rc = gemv_scan_chunk(h, P, qn, b, kk, -1, nullptr, grid, st))) return rc;
  return finish_search(h, P, b, grid, kk, k, kp, qn, out_idx_dev, out_dist_dev, st);
}

int vdb_search_scan_dev(vdb_t* h, const float* queries_dev, int b, int k, int kp, int slot, void* stream) {
  if (!h) return VDB_ERR_ARG;
  if (slot != 0 && slot != 1) return fail(h, VDB_ERR_ARG, "search_scan: slot must be 0 or 1");
  NOT_ON_SHARDED(h, "vdb_search_scan_dev");
  int rc = check_search_args(h, queries_dev, b, k, kp, h, h);
  if (rc) return rc;
  const int kk = kp ? kp : k;
  if (h->split[slot].pending) return fail(h, VDB_ERR_STATE, "search_scan: this slot already holds a pending scan");
  if (b < 1 || b > kQChunk || h->n == 0 || takes_mfma_path(h, b, kk, kp, false))
    return fail(h, VDB_ERR_STATE, "search_scan: this batch does not split (empty store, more than 64 queries, or the batched MFMA path): use vdb_search_dev");
  if ((rc = set_device(h))) return rc;
  hipStream_t st = (hipStream_t)stream;
  float* qn = nullptr;
  if ((rc = prep_queries(h, queries_dev, b, st, &qn, slot))) return rc;
  vdb::Split& sp = h->split[slot];
  sp.ev_on = h->timing_every <= 1 || (h->n_calls % h->timing_every) == 0;
  h->n_calls++;
  if (sp.ev_on) {
    sp.ev = h->ring.data() + 3 * (size_t)(h->n_timed % (int64_t)(h->ring.size() / 3));
    h->n_timed++;
    HIP_TRY(h, hipEventRecord(sp.ev[0], st));
  }
  h->last_bytes = 0.0; h->last_path = 0; h->last_passes = 0;
  const int grid = gemv_grid(h);
  const WsSet W = ws_of(h, slot);
  if ((rc = ensure_parts(h, W, (size_t)b * grid * kk))) return rc;
  if ((rc = gemv_scan_chunk(h, ptrs_of(W), qn, b, kk, -1, nullptr, grid, st))) return rc;
  if (sp.ev_on) HIP_TRY(h, hipEventRecord(sp.ev[1], st));
  sp.pending = true; sp.b = b; sp.k = k; sp.kp = kp; sp.grid = grid;
  return VDB_OK;
}

int vdb_search_finish_dev(vdb_t* h, int slot, int64_t* out_idx_dev, float* out_dist_dev, void* stream) {
  if (!h) return VDB_ERR_ARG;
  if (slot != 0 && slot != 1) return fail(h, VDB_ERR_ARG, "search_finish: slot must be 0 or 1");
  vdb::Split& sp = h->split[slot];
  if (!sp.pending) return fail(h, VDB_ERR_STATE, "search_finish: no scan is pending in this slot");
  if (!out_idx_dev || !out_dist_dev) return fail(h, VDB_ERR_ARG, "search_finish: NULL buffer");
  int rc = set_device(h);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const WsSet W = ws_of(h, slot);
  const int kk = sp.kp ? sp.kp : sp.k;
  sp.pending = false;
  if ((rc = finish_search(h, ptrs_of(W), sp.b, sp.grid, kk, sp.k, sp.kp, (const float*)W.q->p, out_idx_dev, out_dist_dev, st))) return rc;
  if (sp.ev_on) {
    HIP_TRY(h, hipEventRecord(sp.ev[2], st));
    h->ev = sp.ev;
    h->timing_valid = true;
  }
  return VDB_OK;
}

namespace {
// vdb_search and vdb_search_fetch: host queries in, host results out; with out_rows also the stored rows of the hits
// (b x k x dim fp32, zero rows behind a -1 index), gathered on the device right behind the search and brought back in
// the same copy.
int search_host(vdb_t* h, const float* queries_host, int b, int k, int kp, int64_t* out_idx, float* out_dist, float* out_rows) {
  int rc = check_search_args(h, queries_host, b, k, kp, out_idx, out_dist);
  if (rc) return rc;
  if (b == 0) return VDB_OK;
  if ((rc = set_device(h))) return rc;
  const size_t qbytes = (size_t)b * h->dim * 4, ibytes = (size_t)b * k * 8, dbytes = (size_t)b * k * 4;
  const size_t rbytes = out_rows && !is_sharded(h) ? (size_t)b * k * h->dim * 4 : 0;
  if (rbytes > ((size_t)1 << 30)) return fail(h, VDB_ERR_ARG, "search_fetch: more than 1 GiB of rows per call (search, then vdb_fetch in pieces)");
  if ((rc = ensure(h, h->w_stage, qbytes))) return rc;
  if ((rc = ensure(h, h->w_out_i, ibytes + dbytes + rbytes))) return rc;    // [idx | dist | rows] in one buffer: one copy back
  int64_t* d_idx = (int64_t*)h->w_out_i.p;
  float* d_dist = (float*)((char*)h->w_out_i.p + ibytes);
  float* d_rows = (float*)((char*)h->w_out_i.p + ibytes + dbytes);
  // Small calls (every reference call site is batch 1) go through pinned staging: a copy from or to pageable
  // memory is staged by the runtime and costs ~10 us each at any size; three of them were half of a
  // 41 us query against a 14 777-row store.
  const size_t pin_need = qbytes + ibytes + dbytes + rbytes;
  bool pinned = pin_need <= ((size_t)1 << 20);
  if (pinned && h->pin_bytes < pin_need) {
    if (h->pin) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipHostFree(h->pin)); h->pin = nullptr; h->pin_bytes = 0; }
    const size_t want = std::max(pin_need, (size_t)65536);
    if (hipHostMalloc(&h->pin, want, hipHostMallocDefault) == hipSuccess) h->pin_bytes = want;
    else { (void)hipGetLastError(); h->pin = nullptr; pinned = false; }   // no pinned memory to be had: the pageable path still works
  }
  if (pinned) {
    memcpy(h->pin, queries_host, qbytes);
    HIP_TRY(h, hipMemcpyAsync(h->w_stage.p, h->pin, qbytes, hipMemcpyHostToDevice, h->stream));
  } else {
    HIP_TRY(h, hipMemcpyAsync(h->w_stage.p, queries_host, qbytes, hipMemcpyHostToDevice, h->stream));
  }
  if (is_sharded(h)) {
    if ((rc = sharded_search_dev(h, (const float*)h->w_stage.p, b, k, kp, d_idx, d_dist, h->stream))) return rc;
  } else {
    const float* qn = nullptr;
    bool raw = false;
    if ((rc = prep_or_pass(h, (const float*)h->w_stage.p, b, k, kp, false, h->stream, &qn, &raw))) return rc;
    if ((rc = search_core(h, qn, b, k, kp, -1, d_idx, d_dist, h->stream, nullptr, raw))) return rc;
    if (rbytes) {
      const long long nr = (long long)b * k;
      hipLaunchKernelGGL(gather_result_rows_kernel, dim3((int)std::min<long long>((nr + 3) / 4, (long long)h->n_cu * 8)), dim3(256), 0, h->stream,
                         h->keep_f32 ? h->rows_f32 : nullptr, h->rows_bf16, h->ld, h->dim, (const long long*)d_idx, nr,
                         (long long)h->row_offset, d_rows);
      HIP_TRY(h, hipGetLastError());
    }
  }
  if (pinned) {
    char* back = (char*)h->pin + qbytes;
    HIP_TRY(h, hipMemcpyAsync(back, h->w_out_i.p, ibytes + dbytes + rbytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    memcpy(out_idx, back, ibytes);
    memcpy(out_dist, back + ibytes, dbytes);
    if (rbytes) memcpy(out_rows, back + ibytes + dbytes, rbytes);
  } else {
    HIP_TRY(h, hipMemcpyAsync(out_idx, d_idx, ibytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_dist, d_dist, dbytes, hipMemcpyDeviceToHost, h->stream));
    if (rbytes) HIP_TRY(h, hipMemcpyAsync(out_rows, d_rows, rbytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  if (out_rows && is_sharded(h)) {   // the rows live on their shards: fetch them by global index (zero rows behind -1)
    const int64_t nr = (int64_t)b * k;
    std::vector<int64_t> ok;
    std::vector<int64_t> pos;
    for (int64_t j = 0; j < nr; ++j) if (out_idx[j] >= 0) { ok.push_back(out_idx[j]); pos.push_back(j); }
    std::vector<float> tmp(ok.size() * (size_t)h->dim);
    if (!ok.empty() && (rc = vdb_fetch(h, ok.data(), (int64_t)ok.size(), tmp.data()))) return rc;
    memset(out_rows, 0, (size_t)nr * h->dim * 4);
    for (size_t j = 0; j < ok.size(); ++j) memcpy(out_rows + (size_t)pos[j] * h->dim, tmp.data() + j * (size_t)h->dim, (size_t)h->dim * 4);
  }
  return VDB_OK;
}
}  // namespace

int vdb_search(vdb_t* h, const float* queries_host, int b, int k, int kp, int64_t* out_idx, float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  return search_host(h, queries_host, b, k, kp, out_idx, out_dist, nullptr);
}

int vdb_search_fetch(vdb_t* h, const float* queries_host, int b, int k, int kp, int64_t* out_idx, float* out_dist, float* out_rows) {
  if (!h) return VDB_ERR_ARG;
  if (b > 0 && !out_rows) return fail(h, VDB_ERR_ARG, "search_fetch: NULL rows buffer");
  return search_host(h, queries_host, b, k, kp, out_idx, out_dist, out_rows);
}

int vdb_search_masked_dev(vdb_t* h, const float* queries_dev, int b, int k, int kp, const uint32_t* allow_dev,
                          int64_t n_bits, int64_t* out_idx_dev, float* out_dist_dev, void* stream) {
  if (!h) return VDB_ERR_ARG;
  if (!allow_dev && h->n > 0) return fail(h, VDB_ERR_ARG, "search_masked: NULL bitmap");
  if (n_bits != h->n) return fail(h, VDB_ERR_ARG, "search_masked: the bitmap must hold one bit per stored row");
  int rc = check_search_args(h, queries_dev, b, k, kp, out_idx_dev, out_dist_dev);
  if (rc) return rc;
  if (b == 0) return VDB_OK;
  if (is_sharded(h)) {   // the bitmap is over GLOBAL rows (on shard 0's device): every shard takes its word-aligned slice
    ShardedExtra x;
    x.allow_dev0 = allow_dev;
    return sharded_search_dev(h, queries_dev, b, k, kp, out_idx_dev, out_dist_dev, (hipStream_t)stream, x);
  }
  if ((rc = set_device(h))) return rc;
  hipStream_t st = (hipStream_t)stream;
  const float* qn = nullptr;
  bool raw = false;
  if ((rc = prep_or_pass(h, queries_dev, b, k, kp, true, st, &qn, &raw))) return rc;
  return search_core(h, qn, b, k, kp, -1, out_idx_dev, out_dist_dev, st, allow_dev, raw);
}

int vdb_search_masked(vdb_t* h, const float* queries_host, int b, int k, int kp, const uint32_t* allow_host,
                      int64_t n_bits, int64_t* out_idx, float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  if (!allow_host && h->n > 0) return fail(h, VDB_ERR_ARG, "search_masked: NULL bitmap");
  if (n_bits != h->n) return fail(h, VDB_ERR_ARG, "search_masked: the bitmap must hold one bit per stored row");
  int rc = check_search_args(h, queries_host, b, k, kp, out_idx, out_dist);
  if (rc) return rc;
  if (b == 0) return VDB_OK;
  if ((rc = set_device(h))) return rc;
  if (is_sharded(h)) {   // queries to shard 0's device, each shard's slice of the host bitmap straight to its own device
    const size_t ibytes = (size_t)b * k * 8, dbytes = (size_t)b * k * 4;
    if ((rc = ensure(h, h->w_stage, (size_t)b * h->dim * 4))) return rc;
    if ((rc = ensure(h, h->w_out_i, ibytes + dbytes))) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->w_stage.p, queries_host, (size_t)b * h->dim * 4, hipMemcpyHostToDevice, h->stream));
    ShardedExtra x;
    x.allow_host = allow_host;
    if ((rc = sharded_search_dev(h, (const float*)h->w_stage.p, b, k, kp, (int64_t*)h->w_out_i.p, (float*)((char*)h->w_out_i.p + ibytes), h->stream, x))) return rc;
    HIP_TRY(h, hipMemcpyAsync(out_idx, h->w_out_i.p, ibytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(out_dist, (char*)h->w_out_i.p + ibytes, dbytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (vdb_t* c : h->shards) { if ((rc = set_device(c))) return fail(h, rc, c->err); HIP_TRY(h, hipStreamSynchronize(c->stream)); }   // the bitmap copies read the caller's array
    return VDB_OK;
  }
  const size_t words = (size_t)((h->n + 31) / 32);
  if ((rc = ensure(h, h->w_mask, std::max<size_t>(words, 1) * 4))) return rc;
  if ((rc = ensure(h, h->w_stage, (size_t)b * h->dim * 4))) return rc;
  if ((rc = ensure(h, h->w_out_i, (size_t)b * k * 8))) return rc;
  if ((rc = ensure(h, h->w_out_d, (size_t)b * k * 4))) return rc;
  if (words) HIP_TRY(h, hipMemcpyAsync(h->w_mask.p, allow_host, words * 4, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->w_stage.p, queries_host, (size_t)b * h->dim * 4, hipMemcpyHostToDevice, h->stream));
  const float* qn = nullptr;
  bool raw = false;
  if ((rc = prep_or_pass(h, (const float*)h->w_stage.p, b, k, kp, true, h->stream, &qn, &raw))) return rc;
  if ((rc = search_core(h, qn, b, k, kp, -1, (int64_t*)h->w_out_i.p, (float*)h->w_out_d.p, h->stream,
                        (const unsigned*)h->w_mask.p, raw))) return rc;
  HIP_TRY(h, hipMemcpyAsync(out_idx, h->w_out_i.p, (size_t)b * k * 8, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(out_dist, h->w_out_d.p, (size_t)b * k * 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VDB_OK;
}

int vdb_merge_topk_dev(vdb_t* h, const float* cd, int64_t dstride, const int64_t* ci, int64_t istride, int g,
                       int b, int k, int64_t* oi, float* od, void* stream) {
  if (!h) return VDB_ERR_ARG;
  NOT_ON_SHARDED(h, "vdb_merge_topk_dev");
  if (g < 1 || b < 0 || k < 1 || k > VDB_MAX_K) return fail(h, VDB_ERR_ARG, "merge: bad g/b/k");
  if (b > 0 && (!cd || !ci || !oi || !od)) return fail(h, VDB_ERR_ARG, "merge: NULL buffer");
  if (b == 0) return VDB_OK;
  int rc = set_device(h);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream, HIP convention
  if (dstride < 0 || istride < 0) return fail(h, VDB_ERR_ARG, "merge: negative stride");
  hipLaunchKernelGGL(merge_shards_kernel, dim3(b), dim3(256), 0, st, cd, (long long)(dstride ? dstride : (int64_t)b * k),
                     (const long long*)ci, (long long)(istride ? istride : (int64_t)b * k), g, k, (long long*)oi, od);
  HIP_TRY(h, hipGetLastError());
  return VDB_OK;
}

namespace {
// The self-join on a sharded handle (C5 over the GPUs of one process): blocks of at most 4096 rows-as-queries, each
// block cut at shard borders so that it lies in ONE shard (the owner excludes row q0 + q for query q, the others
// exclude nothing).  The owner turns its stored rows into the block's fp32 unit queries, they go to shard 0's device,
// and the block is searched like any batch: every shard scans its rows, P2P stores, one merge.  Results leave through
// two pinned buffers: the host drains block j - 1 while block j computes.
int sharded_selfjoin(vdb_t* h, int64_t q_row0, int64_t q_n, int k, int exclude_self, int kp, int64_t* out_idx, float* out_dist) {
  int rc;
  const int qblock = kMfMaxQueriesPerCall;
  const size_t ibytes = (size_t)qblock * k * 8, dbytes = (size_t)qblock * k * 4, obytes = ibytes + dbytes;
  if ((rc = ensure(h, h->w_q, (size_t)qblock * h->ld * 4))) return rc;
  if ((rc = ensure(h, h->w_out_i, 2 * obytes))) return rc;
  if (h->pin_bytes < 2 * obytes) {
    if (h->pin) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipHostFree(h->pin)); h->pin = nullptr; h->pin_bytes = 0; }
    HIP_TRY(h, hipHostMalloc(&h->pin, 2 * obytes, hipHostMallocDefault));
    h->pin_bytes = 2 * obytes;
  }
  hipEvent_t copied[2] = {nullptr, nullptr};   // (created with shard 0's device current: they are recorded on h->stream only)
  auto cleanup = [&]() { for (hipEvent_t e : copied) if (e) (void)hipEventDestroy(e); };
  if (hipEventCreateWithFlags(&copied[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&copied[1], hipEventDisableTiming) != hipSuccess) {
    cleanup();
    return fail(h, VDB_ERR_DEVICE, "selfjoin: event creation failed");
  }
  auto drain = [&](int64_t q0, int cb, int slot) -> int {
    HIP_TRY(h, hipEventSynchronize(copied[slot]));
    const char* src = (const char*)h->pin + (size_t)slot * obytes;
    memcpy(out_idx + (size_t)(q0 - q_row0) * k, src, (size_t)cb * k * 8);
    memcpy(out_dist + (size_t)(q0 - q_row0) * k, src + ibytes, (size_t)cb * k * 4);
    return VDB_OK;
  };
  int64_t prev_q0 = -1; int prev_cb = 0, prev_slot = 0, slot = 0;
  rc = VDB_OK;
  for (int64_t q0 = q_row0; q0 < q_row0 + q_n && rc == VDB_OK; slot ^= 1) {
    const ShardSpan sp = shard_span(h, q0, std::min<int64_t>(qblock, q_row0 + q_n - q0));
    const int cb = (int)sp.count;
    vdb_t* o = h->shards[(size_t)sp.s];
    int64_t* d_idx = (int64_t*)((char*)h->w_out_i.p + (size_t)slot * obytes);
    float* d_dist = (float*)((char*)d_idx + ibytes);
    auto body = [&]() -> int {
      int r2;
      // the owner's rows as fp32 unit queries [cb][ld], on the owner's device, then on shard 0's
      if ((r2 = set_device(o))) return r2;
      if ((r2 = ensure(o, o->w_q1, (size_t)qblock * h->ld * 4))) return r2;
      HIP_TRY(h, hipStreamWaitEvent(o->stream, copied[slot ^ 1], 0));   // (no-op before the first record) the previous block's searches have read the query buffer
      hipLaunchKernelGGL(rows_to_f32_kernel, dim3(std::min(4096, (cb * h->ld + 255) / 256)), dim3(256), 0, o->stream,
                         o->keep_f32 ? o->rows_f32 : nullptr, o->rows_bf16, h->ld, (long long)(q0 - o->row_offset), cb, (float*)o->w_q1.p);
      HIP_TRY(h, hipGetLastError());
      if (o->device != h->device) HIP_TRY(h, hipMemcpyPeerAsync(h->w_q.p, h->device, o->w_q1.p, o->device, (size_t)cb * h->ld * 4, o->stream));
      else HIP_TRY(h, hipMemcpyAsync(h->w_q.p, o->w_q1.p, (size_t)cb * h->ld * 4, hipMemcpyDeviceToDevice, o->stream));
      hipEvent_t q_ready = h->shard_aux[(size_t)sp.s];   // the owner's own event: recorded on the owner's stream
      HIP_TRY(h, hipEventRecord(q_ready, o->stream));
      if ((r2 = set_device(h))) return r2;
      HIP_TRY(h, hipStreamWaitEvent(h->stream, q_ready, 0));
      ShardedExtra x;
      x.q_unit_ld = true;
      x.self_global0 = exclude_self ? q0 : -1;
      if ((r2 = sharded_search_dev(h, (const float*)h->w_q.p, cb, k, kp, d_idx, d_dist, h->stream, x))) return r2;
      char* dst = (char*)h->pin + (size_t)slot * obytes;
      HIP_TRY(h, hipMemcpyAsync(dst, d_idx, (size_t)cb * k * 8, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipMemcpyAsync(dst + ibytes, d_dist, (size_t)cb * k * 4, hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipEventRecord(copied[slot], h->stream));
      return VDB_OK;
    };
    rc = body();
    if (rc == VDB_OK && prev_q0 >= 0) rc = drain(prev_q0, prev_cb, prev_slot);
    prev_q0 = q0; prev_cb = cb; prev_slot = slot;
    q0 += cb;
  }
  if (rc == VDB_OK && prev_q0 >= 0) rc = drain(prev_q0, prev_cb, prev_slot);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (vdb_t* c : h->shards) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
  (void)hipSetDevice(h->device);
  cleanup();
  return rc;
}
}  // namespace

int vdb_selfjoin(vdb_t* h, int64_t q_row0, int64_t q_n, int k, int exclude_self, int kp, int64_t* out_idx,
                 float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  if (q_row0 < 0 || q_n < 0 || q_row0 + q_n > h->n) return fail(h, VDB_ERR_ARG, "selfjoin: query range outside the store");
  int rc = check_search_args(h, h, q_n > 0 ? 1 : 0, k, kp, out_idx, out_dist);
  if (rc) return rc;
  if (q_n == 0) return VDB_OK;
  if ((rc = set_device(h))) return rc;
  if (is_sharded(h)) return sharded_selfjoin(h, q_row0, q_n, k, exclude_self, kp, out_idx, out_dist);
  // Blocks of 4096 rows-as-queries.  The device never waits for the host: block j's results go to one of two
  // device buffers, a copy stream brings them to pinned memory while block j+1 computes, and the host moves
  // block j-1 into the caller's arrays meanwhile.
  const int qblock = kMfMaxQueriesPerCall;
  const size_t ibytes = (size_t)qblock * k * 8, dbytes = (size_t)qblock * k * 4, obytes = ibytes + dbytes;
  if ((rc = ensure(h, h->w_q, (size_t)qblock * h->ld * 4))) return rc;
  if ((rc = ensure(h, h->w_out_i, 2 * obytes))) return rc;
  if (h->pin_bytes < 2 * obytes) {
    if (h->pin) { HIP_TRY(h, hipStreamSynchronize(h->stream)); HIP_TRY(h, hipHostFree(h->pin)); h->pin = nullptr; h->pin_bytes = 0; }
    HIP_TRY(h, hipHostMalloc(&h->pin, 2 * obytes, hipHostMallocDefault));
    h->pin_bytes = 2 * obytes;
  }
  if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  hipEvent_t computed[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
  auto cleanup = [&]() { for (int i = 0; i < 2; ++i) { if (computed[i]) (void)hipEventDestroy(computed[i]); if (copied[i]) (void)hipEventDestroy(copied[i]); } };
  for (int i = 0; i < 2; ++i)
    if (hipEventCreateWithFlags(&computed[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&copied[i], hipEventDisableTiming) != hipSuccess) {
      cleanup();
      return fail(h, VDB_ERR_DEVICE, "selfjoin: event creation failed");
    }
  auto drain = [&](int64_t q0, int cb, int slot) -> int {   // host side of block (q0, cb): pinned -> caller
    HIP_TRY(h, hipEventSynchronize(copied[slot]));
    const char* src = (const char*)h->pin + (size_t)slot * obytes;
    memcpy(out_idx + (size_t)q0 * k, src, (size_t)cb * k * 8);
    memcpy(out_dist + (size_t)q0 * k, src + ibytes, (size_t)cb * k * 4);
    return VDB_OK;
  };
  int64_t prev_q0 = -1; int prev_cb = 0, prev_slot = 0;
  int slot = 0;
  rc = VDB_OK;
  for (int64_t q0 = 0; q0 < q_n && rc == VDB_OK; q0 += qblock, slot ^= 1) {
    const int cb = (int)std::min<int64_t>(qblock, q_n - q0);
    int64_t* d_idx = (int64_t*)((char*)h->w_out_i.p + (size_t)slot * obytes);
    float* d_dist = (float*)((char*)d_idx + ibytes);
    auto body = [&]() -> int {
      // this slot's previous copy (two blocks ago) was drained by the host before block q0 - qblock was issued
      // stored rows are unit vectors already: copy them as the normalised queries
      hipLaunchKernelGGL(rows_to_f32_kernel, dim3(std::min(4096, (cb * h->ld + 255) / 256)), dim3(256), 0, h->stream,
                         h->keep_f32 ? h->rows_f32 : nullptr, h->rows_bf16, h->ld, (long long)(q_row0 + q0), cb, (float*)h->w_q.p);
      HIP_TRY(h, hipGetLastError());
      int r2;
      if ((r2 = search_core(h, (const float*)h->w_q.p, cb, k, kp, exclude_self ? q_row0 + q0 : -1, d_idx, d_dist, h->stream))) return r2;
      HIP_TRY(h, hipEventRecord(computed[slot], h->stream));
      HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, computed[slot], 0));
      char* dst = (char*)h->pin + (size_t)slot * obytes;
      HIP_TRY(h, hipMemcpyAsync(dst, d_idx, (size_t)cb * k * 8, hipMemcpyDeviceToHost, h->copy_stream));
      HIP_TRY(h, hipMemcpyAsync(dst + ibytes, d_dist, (size_t)cb * k * 4, hipMemcpyDeviceToHost, h->copy_stream));
      HIP_TRY(h, hipEventRecord(copied[slot], h->copy_stream));
      return VDB_OK;
    };
    rc = body();
    if (rc == VDB_OK && prev_q0 >= 0) rc = drain(prev_q0, prev_cb, prev_slot);   // overlaps this block's compute
    prev_q0 = q0; prev_cb = cb; prev_slot = slot;
  }
  if (rc == VDB_OK && prev_q0 >= 0) rc = drain(prev_q0, prev_cb, prev_slot);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamSynchronize(h->copy_stream);
  cleanup();
  return rc;
}

int vdb_fetch(vdb_t* h, const int64_t* idx_host, int64_t n, float* out_rows_host) {
  if (!h) return VDB_ERR_ARG;
  if (n < 0 || (n > 0 && (!idx_host || !out_rows_host))) return fail(h, VDB_ERR_ARG, "fetch: NULL buffer");
  for (int64_t j = 0; j < n; ++j)
    if (idx_host[j] < 0 || idx_host[j] >= h->n) return fail(h, VDB_ERR_ARG, "fetch: index out of range");
  if (n == 0) return VDB_OK;
  if (is_sharded(h)) {   // route each global index to its shard (runs of one shard go down together)
    std::vector<int64_t> local;
    for (int64_t j = 0; j < n;) {
      const int sh = shard_span(h, idx_host[j], 1).s;
      local.clear();
      int64_t e = j;
      while (e < n && shard_span(h, idx_host[e], 1).s == sh) { local.push_back(idx_host[e] - (int64_t)sh * h->per); ++e; }
      vdb_t* c = h->shards[(size_t)sh];
      const int rc = vdb_fetch(c, local.data(), (int64_t)local.size(), out_rows_host + (size_t)j * h->dim);
      if (rc) return fail(h, rc, c->err);
      j = e;
    }
    return VDB_OK;
  }
  int rc = set_device(h);
  if (rc) return rc;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)h->dim * 4));
  if ((rc = ensure(h, h->w_idx, (size_t)std::min(n, chunk) * 8))) return rc;
  if ((rc = ensure(h, h->w_stage, (size_t)std::min(n, chunk) * h->dim * 4))) return rc;
  for (int64_t r = 0; r < n; r += chunk) {
    const int64_t m = std::min(chunk, n - r);
    HIP_TRY(h, hipMemcpyAsync(h->w_idx.p, idx_host + r, (size_t)m * 8, hipMemcpyHostToDevice, h->stream));
    const int blocks = (int)std::min<int64_t>((m + 3) / 4, (int64_t)h->n_cu * 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, h->stream, h->keep_f32 ? h->rows_f32 : nullptr,
                       h->rows_bf16, h->ld, h->dim, (const long long*)h->w_idx.p, (long long)m, (float*)h->w_stage.p);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(out_rows_host + (size_t)r * h->dim, h->w_stage.p, (size_t)m * h->dim * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return VDB_OK;
}

// ---- persistence: 32-byte header + unpadded rows (shared with the oracle) --
struct vdb_file_header { char magic[8]; int32_t dim; int32_t stored_dtype; int64_t n; int64_t row_offset; };

namespace {
// append the stored rows of one (unsharded) handle to an open file
int write_rows(vdb_t* h, FILE* f, std::string* why) {
  int rc = set_device(h);
  if (rc) { *why = h->err; return rc; }
  const size_t esz = h->keep_f32 ? 4 : 2;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)h->dim * 4));
  std::vector<unsigned char> host((size_t)std::min<int64_t>(std::max<int64_t>(h->n, 1), chunk) * h->dim * esz);
  for (int64_t r = 0; r < h->n; r += chunk) {
    const int64_t m = std::min(chunk, h->n - r);
    const unsigned char* src = h->keep_f32 ? (const unsigned char*)(h->rows_f32 + (size_t)r * h->ld)
                                           : (const unsigned char*)(h->rows_bf16 + (size_t)r * h->ld);
    hipError_t e = hipMemcpy2DAsync(host.data(), (size_t)h->dim * esz, src, (size_t)h->ld * esz, (size_t)h->dim * esz, (size_t)m, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { *why = std::string("save: ") + hipGetErrorString(e); return VDB_ERR_DEVICE; }
    if (fwrite(host.data(), esz, (size_t)m * h->dim, f) != (size_t)m * h->dim) { *why = "save: short write"; return VDB_ERR_IO; }
  }
  return VDB_OK;
}

// read rows [0, n) of the open file (positioned at their first byte) into an (unsharded) handle
int read_rows(vdb_t* h, FILE* f, int stored_dtype, int64_t n, std::string* why) {
  int rc = set_device(h);
  if (rc) { *why = h->err; return rc; }
  if ((rc = grow(h, std::max<int64_t>(n, 1)))) { *why = h->err; return rc; }
  const int dim = h->dim;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(64u << 20) / ((int64_t)dim * 4));
  if (stored_dtype == VDB_F32) {
    // rows on file are unit vectors; re-normalising is idempotent to <= 1 ulp, so
    // the verbatim bits are kept for the fp32 copy and bf16 is derived from them.
    std::vector<float> host((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), chunk) * dim);
    if ((rc = ensure(h, h->w_stage, host.size() * 4))) { *why = h->err; return rc; }
    for (int64_t r = 0; r < n; r += chunk) {
      const int64_t m = std::min(chunk, n - r);
      if (fread(host.data(), 4, (size_t)m * dim, f) != (size_t)m * dim) { *why = "load: short read"; return VDB_ERR_IO; }
      hipError_t e = hipMemcpyAsync(h->w_stage.p, host.data(), (size_t)m * dim * 4, hipMemcpyHostToDevice, h->stream);
      if (e != hipSuccess) { *why = hipGetErrorString(e); return VDB_ERR_DEVICE; }
      hipLaunchKernelGGL(copy_rows_kernel, dim3((int)std::min<int64_t>((m + 3) / 4, (int64_t)h->n_cu * 8)), dim3(256), 0, h->stream,
                         (const float*)h->w_stage.p, (long long)m, dim, h->ld,
                         h->keep_f32 ? h->rows_f32 + (size_t)r * h->ld : nullptr,
                         h->dtype == VDB_BF16 ? h->rows_bf16 + (size_t)r * h->ld : nullptr);
      e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) { *why = hipGetErrorString(e); return VDB_ERR_DEVICE; }
    }
  } else {
    std::vector<uint16_t> host((size_t)std::min<int64_t>(std::max<int64_t>(n, 1), chunk) * dim);
    for (int64_t r = 0; r < n; r += chunk) {
      const int64_t m = std::min(chunk, n - r);
      if (fread(host.data(), 2, (size_t)m * dim, f) != (size_t)m * dim) { *why = "load: short read"; return VDB_ERR_IO; }
      hipError_t e = hipMemset2DAsync(h->rows_bf16 + (size_t)r * h->ld, (size_t)h->ld * 2, 0, (size_t)h->ld * 2, (size_t)m, h->stream);
      if (e == hipSuccess) e = hipMemcpy2DAsync(h->rows_bf16 + (size_t)r * h->ld, (size_t)h->ld * 2, host.data(), (size_t)dim * 2, (size_t)dim * 2, (size_t)m, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      if (e != hipSuccess) { *why = hipGetErrorString(e); return VDB_ERR_DEVICE; }
    }
  }
  h->n = n;
  return VDB_OK;
}

// open <dir>/vectors.vdb and validate its header against the file's length
int open_store(const char* dir, FILE** out, vdb_file_header* hd) {
  const std::string path = std::string(dir) + "/vectors.vdb";
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return fail(nullptr, VDB_ERR_IO, "load: cannot open " + path);
  if (fread(hd, sizeof *hd, 1, f) != 1 || memcmp(hd->magic, "VDBFLAT1", 8) != 0) { fclose(f); return fail(nullptr, VDB_ERR_IO, "load: bad header"); }
  if (hd->dim <= 0 || hd->dim > 32768 || (hd->stored_dtype != VDB_F32 && hd->stored_dtype != VDB_BF16) || hd->n < 0) {
    fclose(f);
    return fail(nullptr, VDB_ERR_IO, "load: bad header (dim, dtype or row count out of range)");
  }
  struct stat sb;
  const int64_t want = (int64_t)sizeof *hd + hd->n * (int64_t)hd->dim * (hd->stored_dtype == VDB_F32 ? 4 : 2);
  if (fstat(fileno(f), &sb) != 0 || (int64_t)sb.st_size != want) {
    fclose(f);
    return fail(nullptr, VDB_ERR_IO, "load: " + path + " is truncated or its header disagrees with its length");
  }
  *out = f;
  return VDB_OK;
}
}  // namespace

// One flat file whatever the handle: 32-byte header + the rows in global order (a sharded handle writes its shards one
// after the other), written to a temporary name and renamed into place, so a crash never leaves a half-written store.
int vdb_save(vdb_t* h, const char* dir) {
  if (!h || !dir) return VDB_ERR_ARG;
  if (mkdir(dir, 0777) != 0 && errno != EEXIST) return fail(h, VDB_ERR_IO, std::string("save: cannot create ") + dir);
  const std::string path = std::string(dir) + "/vectors.vdb", tmp = path + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return fail(h, VDB_ERR_IO, "save: cannot open " + tmp);
  vdb_file_header hd;
  memset(&hd, 0, sizeof hd);
  memcpy(hd.magic, "VDBFLAT1", 8);
  hd.dim = h->dim; hd.stored_dtype = h->keep_f32 ? VDB_F32 : VDB_BF16; hd.n = h->n; hd.row_offset = h->row_offset;
  int rc = fwrite(&hd, sizeof hd, 1, f) == 1 ? VDB_OK : VDB_ERR_IO;
  std::string why = "save: short write";
  if (rc == VDB_OK) {
    if (is_sharded(h)) { for (vdb_t* c : h->shards) if ((rc = write_rows(c, f, &why))) break; }
    else rc = write_rows(h, f, &why);
  }
  if (rc == VDB_OK && (fflush(f) != 0 || fsync(fileno(f)) != 0)) rc = VDB_ERR_IO;
  fclose(f);
  if (rc == VDB_OK && rename(tmp.c_str(), path.c_str()) != 0) { rc = VDB_ERR_IO; why = "save: cannot rename " + tmp; }
  if (rc != VDB_OK) { unlink(tmp.c_str()); return fail(h, rc, why); }
  return VDB_OK;
}

int vdb_load(vdb_t** out, const char* dir, int dtype, int keep_f32, int device_id) {
  if (!out || !dir) return fail(nullptr, VDB_ERR_ARG, "load: NULL argument");
  FILE* f = nullptr;
  vdb_file_header hd;
  int rc = open_store(dir, &f, &hd);
  if (rc) return rc;
  if (hd.n > (int64_t)0x7ffffff0) { fclose(f); return fail(nullptr, VDB_ERR_IO, "load: more rows than one shard holds (use vdb_load_sharded)"); }
  if (hd.stored_dtype == VDB_BF16 && (dtype == VDB_F32 || keep_f32)) { fclose(f); return fail(nullptr, VDB_ERR_STATE, "load: file holds bf16 rows only; cannot rebuild fp32 rows"); }
  vdb_t* h = nullptr;
  rc = vdb_create(&h, hd.dim, dtype, keep_f32, device_id);
  if (rc) { fclose(f); return rc; }
  h->row_offset = hd.row_offset;
  std::string why;
  rc = read_rows(h, f, hd.stored_dtype, hd.n, &why);
  fclose(f);
  if (rc) { vdb_destroy(h); return fail(nullptr, rc, why); }
  *out = h;
  return VDB_OK;
}

int vdb_load_sharded(vdb_t** out, const char* dir, int dtype, int keep_f32, const int* device_ids, int n_devices) {
  if (!out || !dir) return fail(nullptr, VDB_ERR_ARG, "load: NULL argument");
  FILE* f = nullptr;
  vdb_file_header hd;
  int rc = open_store(dir, &f, &hd);
  if (rc) return rc;
  if (hd.stored_dtype == VDB_BF16 && (dtype == VDB_F32 || keep_f32)) { fclose(f); return fail(nullptr, VDB_ERR_STATE, "load: file holds bf16 rows only; cannot rebuild fp32 rows"); }
  vdb_t* p = nullptr;
  rc = vdb_create_sharded(&p, hd.dim, dtype, keep_f32, device_ids, n_devices);
  if (rc) { fclose(f); return rc; }
  std::string why;
  if ((rc = vdb_reserve(p, hd.n))) why = p->err;
  for (size_t s = 0; rc == VDB_OK && s < p->shards.size(); ++s) {
    const int64_t lo = std::min<int64_t>(hd.n, (int64_t)s * p->per);
    const int64_t hi = s + 1 == p->shards.size() ? hd.n : std::min<int64_t>(hd.n, (int64_t)(s + 1) * p->per);
    rc = read_rows(p->shards[s], f, hd.stored_dtype, hi - lo, &why);
  }
  fclose(f);
  if (rc) { vdb_destroy(p); return fail(nullptr, rc, why); }
  p->n = hd.n;
  *out = p;
  return VDB_OK;
}

int vdb_last_timing(vdb_t* h, float* scan_ms, float* merge_ms, double* bytes) {
  if (!h) return VDB_ERR_ARG;
  if (is_sharded(h)) {   // the slowest shard's phases; bytes summed over the shards
    float a = 0.f, b = 0.f; double by = 0.0;
    for (vdb_t* c : h->shards) {
      float x = 0.f, y = 0.f; double z = 0.0;
      const int rc = vdb_last_timing(c, &x, &y, &z);
      if (rc) return fail(h, rc, c->err);
      a = std::max(a, x); b = std::max(b, y); by += z;
    }
    if (scan_ms) *scan_ms = a;
    if (merge_ms) *merge_ms = b;
    if (bytes) *bytes = by;
    return VDB_OK;
  }
  if (!h->timing_valid) return fail(h, VDB_ERR_STATE, "last_timing: no search has run on this handle");
  int rc = set_device(h);
  if (rc) return rc;
  HIP_TRY(h, hipEventSynchronize(h->ev[2]));
  float a = 0.f, b = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
  HIP_TRY(h, hipEventElapsedTime(&b, h->ev[1], h->ev[2]));
  if (scan_ms) *scan_ms = a;
  if (merge_ms) *merge_ms = b;
  if (bytes) *bytes = h->last_bytes;
  return VDB_OK;
}

int vdb_timing_stats(vdb_t* h, int64_t* n_searches, double* scan_ms_sum, double* merge_ms_sum, int reset) {
  if (!h) return VDB_ERR_ARG;
  if (is_sharded(h)) {   // the shard with the largest scan total (every shard records the same searches)
    int64_t n = 0; double a = 0.0, b = 0.0;
    for (vdb_t* c : h->shards) {
      int64_t cn = 0; double x = 0.0, y = 0.0;
      const int rc = vdb_timing_stats(c, &cn, &x, &y, reset);
      if (rc) return fail(h, rc, c->err);
      if (x >= a) { a = x; b = y; n = cn; }
    }
    if (n_searches) *n_searches = n;
    if (scan_ms_sum) *scan_ms_sum = a;
    if (merge_ms_sum) *merge_ms_sum = b;
    return VDB_OK;
  }
  int rc = set_device(h);
  if (rc) return rc;
  const int64_t cap = (int64_t)(h->ring.size() / 3);
  const int64_t cnt = std::min(h->n_timed, cap);
  double a = 0.0, b = 0.0;
  for (int64_t i = 0; i < cnt; ++i) {
    hipEvent_t* e = h->ring.data() + 3 * (size_t)i;
    float x = 0.f, y = 0.f;
    HIP_TRY(h, hipEventSynchronize(e[2]));
    HIP_TRY(h, hipEventElapsedTime(&x, e[0], e[1]));
    HIP_TRY(h, hipEventElapsedTime(&y, e[1], e[2]));
    a += x; b += y;
  }
  if (n_searches) *n_searches = cnt;
  if (scan_ms_sum) *scan_ms_sum = a;
  if (merge_ms_sum) *merge_ms_sum = b;
  if (reset) h->n_timed = 0;
  return VDB_OK;
}

int vdb_set_option(vdb_t* h, const char* name, int64_t value) {
  if (!h || !name) return VDB_ERR_ARG;
  if (is_sharded(h) && !strcmp(name, "shard_exchange_rccl")) {
    // 1: the shards' k results are exchanged by ONE RCCL all-gather (every shard receives all of them; shard 0 merges);
    // 0 (default): each shard stores its results into shard 0's gather buffer (P2P)
    if (!value) { for (void* c : h->comms) if (c) (void)g_rccl.CommDestroy(c); h->comms.clear(); return VDB_OK; }
    if (!h->comms.empty()) return VDB_OK;
    const int g = (int)h->shards.size();
    std::vector<int> devs;
    for (vdb_t* c : h->shards) devs.push_back(c->device);
    for (int i = 0; i < g; ++i) for (int j = 0; j < i; ++j)
      if (devs[i] == devs[j]) return fail(h, VDB_ERR_STATE, "shard_exchange_rccl: RCCL takes one rank per GPU; this handle has two shards on one device");
    std::string why;
    if (!load_rccl(&why)) return fail(h, VDB_ERR_STATE, why);
    std::vector<void*> comms((size_t)g, nullptr);
    const int rc = g_rccl.CommInitAll(comms.data(), g, devs.data());
    if (rc != 0) return fail(h, VDB_ERR_DEVICE, "ncclCommInitAll: " + rccl_err(rc));
    h->comms = comms;
    return set_device(h);
  }
  if (is_sharded(h) && !strcmp(name, "shard_force_remote")) { h->shard_force_remote = value != 0; return VDB_OK; }
  if (is_sharded(h) && !strcmp(name, "shard_own_stream_for_shard0")) { h->shard_serial_streams = value != 0; return VDB_OK; }
  if (is_sharded(h)) {   // every shard gets the same setting
    for (vdb_t* c : h->shards) { const int rc = vdb_set_option(c, name, value); if (rc) return fail(h, rc, c->err); }
    return VDB_OK;
  }
  if (!strcmp(name, "scan_blocks_per_cu")) { if (value < 1 || value > 32) return fail(h, VDB_ERR_ARG, "scan_blocks_per_cu must be in [1,32]"); h->scan_blocks_per_cu = (int)value; return VDB_OK; }
  if (!strcmp(name, "timing_ring")) {
    if (value < 1 || value > 65536) return fail(h, VDB_ERR_ARG, "timing_ring must be in [1,65536]");
    int rc = set_device(h);
    if (rc) return rc;
    if (h->split[0].pending || h->split[1].pending) return fail(h, VDB_ERR_STATE, "timing_ring: a two-phase scan holds events of the current ring");
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return resize_ring(h, (int)value);
  }
  if (!strcmp(name, "timing_every")) {
    if (value < 1 || value > 65536) return fail(h, VDB_ERR_ARG, "timing_every must be in [1,65536]");
    h->timing_every = (int)value; h->n_calls = 0; return VDB_OK;
  }
  if (!strcmp(name, "scan_nt_loads")) { h->scan_nt_loads = value != 0; return VDB_OK; }
  if (!strcmp(name, "scan_group_launch")) { h->scan_group_launch = value != 0; return VDB_OK; }
  if (!strcmp(name, "scan_xcd_slices")) { h->scan_xcd_slices = value != 0; return VDB_OK; }
  if (!strcmp(name, "scan_super_rounds")) {   // tiles per wave of the grid and super-round (1 = the default, for the on/off form; 0 = off)
    if (value < 0 || value > 64) return fail(h, VDB_ERR_ARG, "scan_super_rounds must be in [0,64]");
    h->scan_super_rounds = value == 1 ? kFusedTilesPerWaveAndRoundC : (int)value; return VDB_OK;
  }
  if (!strcmp(name, "alloc_f32_first")) { h->alloc_f32_first = value != 0; return VDB_OK; }
  if (!strcmp(name, "stream_nt_loads")) { h->stream_nt = value != 0; return VDB_OK; }
  if (!strcmp(name, "stream_xcd_slices")) { h->stream_xcd_slices = value != 0; return VDB_OK; }
  if (!strcmp(name, "mfma_ablate")) { h->mfma_ablate = (int)value; return VDB_OK; }
  if (!strcmp(name, "mfma_variant")) {
    if (value < 4 || value > 6) return fail(h, VDB_ERR_ARG, "mfma_variant must be 4, 5 or 6");
    h->mfma_variant = (int)value; return VDB_OK;
  }
  if (!strcmp(name, "mfma_split_x1000")) {
    if (value != 0 && (value < 100 || value > 900)) return fail(h, VDB_ERR_ARG, "mfma_split_x1000 must be 0 (by kernel family) or in [100,900]");
    h->mfma_split_x1000 = (int)value; return VDB_OK;
  }
  if (!strcmp(name, "mfma_skip_sample")) { h->mfma_skip_sample = value != 0; return VDB_OK; }
  if (!strcmp(name, "mfma_one_pass_span")) { h->mfma_one_pass_span = (int)std::max<int64_t>(1, std::min<int64_t>(value, 1 << 20)); return VDB_OK; }
  if (!strcmp(name, "mfma_dense_tiles")) { if (value < 0 || value > 128) return fail(h, VDB_ERR_ARG, "mfma_dense_tiles must be 0 (by store size) or in [1,128]"); h->mfma_dense_tiles = (int)value; return VDB_OK; }
  if (!strcmp(name, "mfma_force_fallback")) { h->mfma_force_fallback = value != 0; return VDB_OK; }
  if (!strcmp(name, "mfma_append_cap")) {
    if (value < 256 || value > (1 << 22)) return fail(h, VDB_ERR_ARG, "mfma_append_cap must be in [256, 4194304]");
    h->mfma_append_cap = (int)value; return VDB_OK;
  }
  if (!strcmp(name, "mfma_cost_model")) { h->mfma_cost_model = value != 0; return VDB_OK; }
  if (!strcmp(name, "fuse_small_batch")) { h->fuse_small = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return VDB_OK; }   // 0 never, 1 where it wins, 2 wherever the shape allows
  if (!strcmp(name, "scan_balance")) { h->scan_balance = (int)std::max<int64_t>(0, std::min<int64_t>(value, 2)); return VDB_OK; }   // 0: every wave of the grid takes every nw-th tile
  if (!strcmp(name, "scan_stamps")) {   // diagnostic: the fused kernel writes clock stamps (read back with get_stat "scan_stamp_<i>")
    int rc = set_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipDeviceSynchronize());
    if (!value) { if (h->w_stamps.p) { HIP_TRY(h, hipFree(h->w_stamps.p)); h->w_stamps.p = nullptr; h->w_stamps.bytes = 0; } return VDB_OK; }
    if ((rc = ensure(h, h->w_stamps, (size_t)(16 + 6 * kFusedMaxParts) * 8))) return rc;
    HIP_TRY(h, hipMemset(h->w_stamps.p, 0, h->w_stamps.bytes));
    return VDB_OK;
  }
  if (!strcmp(name, "batch_mfma_min_b")) { if (value < 1) return fail(h, VDB_ERR_ARG, "batch_mfma_min_b must be >= 1"); h->batch_mfma_min_b = (int)std::min<int64_t>(value, 1 << 30); return VDB_OK; }
  return fail(h, VDB_ERR_ARG, std::string("set_option: unknown option ") + name);
}

int vdb_get_stat(vdb_t* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return VDB_ERR_ARG;
  if (!strcmp(name, "rows")) { *value = h->n; return VDB_OK; }
  if (!strcmp(name, "n_shards")) { *value = is_sharded(h) ? (int64_t)h->shards.size() : 1; return VDB_OK; }
  if (!strcmp(name, "scratch_bytes")) { *value = (int64_t)kNaScratchBytes; return VDB_OK; }
  if (!strcmp(name, "workspace_allocs")) { *value = h->n_ws_allocs; for (vdb_t* c : h->shards) *value += c->n_ws_allocs; return VDB_OK; }
  if (is_sharded(h)) {
    if (!strcmp(name, "rows_per_shard")) { *value = h->per; return VDB_OK; }
    if (!strcmp(name, "capacity_rows")) { *value = 0; for (vdb_t* c : h->shards) *value += c->cap; return VDB_OK; }
    if (!strcmp(name, "shard_exchange_rccl")) { *value = h->comms.empty() ? 0 : 1; return VDB_OK; }
    if (!strcmp(name, "p2p_direct_shards")) { *value = 0; for (char d : h->shard_direct) *value += d; return VDB_OK; }
    if (!strncmp(name, "sharded_issue_phase_", 20) && name[20] >= '0' && name[20] <= '7') {   // ns per call since the last read of sharded_issue_ns_avg
      *value = h->issue_calls ? h->issue_phase_ns[name[20] - '0'] / h->issue_calls : 0;
      return VDB_OK;
    }
    if (!strcmp(name, "sharded_issue_ns_avg")) {
      for (int64_t& v : h->issue_phase_ns) v = 0;   // mean host time per sharded search since the last read
      *value = h->issue_calls ? h->issue_ns_sum / h->issue_calls : 0;
      h->issue_ns_sum = 0; h->issue_calls = 0;
      return VDB_OK;
    }
    const int rc = vdb_get_stat(h->shards[0], name, value);   // everything else: shard 0's
    return rc ? fail(h, rc, h->shards[0]->err) : VDB_OK;
  }
  if (!strcmp(name, "capacity_rows")) { *value = h->cap; return VDB_OK; }
  if (!strcmp(name, "last_path")) { *value = h->last_path; return VDB_OK; }
  if (!strcmp(name, "last_corpus_passes")) { *value = h->last_passes; return VDB_OK; }
  if (!strcmp(name, "mfma_last_overflow")) {
    *value = 0;
    if (!h->w_cand_cnt.p || h->mfma_last_b <= 0) return VDB_OK;
    int rc = set_device(h);
    if (rc) return rc;
    int flag = 0;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(&flag, (int*)h->w_cand_cnt.p + (size_t)h->mfma_last_b * kMfCntStrideC, 4, hipMemcpyDeviceToHost));
    *value = flag != 0;
    return VDB_OK;
  }
  if (!strncmp(name, "mfma_prof_xcd_", 14) && name[14] >= '0' && name[14] <= '7') {   // diagnostic (mfma_ablate=8): mean cycles per wave (waits + rest) of the blocks of one XCD in the last pass
    *value = 0;
    if (!h->w_dense.p) return VDB_OK;
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> pr((size_t)h->n_cu * 8 * 4);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(pr.data(), h->w_dense.p, pr.size() * 4, hipMemcpyDeviceToHost));
    double tot = 0; size_t cnt = 0;
    for (size_t i = 0; i < pr.size() / 4; ++i)
      if (((i / 8) & 7) == (size_t)(name[14] - '0')) { tot += (double)pr[i * 4] + pr[i * 4 + 1] + pr[i * 4 + 2]; ++cnt; }
    *value = cnt ? (int64_t)(tot / (double)cnt) : 0;
    return VDB_OK;
  }
  if (!strncmp(name, "mfma_prof_", 10) && name[10] >= 'a' && name[10] <= 'd') {   // diagnostic: mean stamp totals of mfma_ablate=8
    *value = 0;
    if (!h->w_dense.p) return VDB_OK;
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<float> pr((size_t)h->n_cu * 8 * 4);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(pr.data(), h->w_dense.p, pr.size() * 4, hipMemcpyDeviceToHost));
    double tot = 0;
    for (size_t i = 0; i < pr.size() / 4; ++i) tot += pr[i * 4 + (name[10] - 'a')];
    *value = (int64_t)(tot / (pr.size() / 4));
    return VDB_OK;
  }
  if (!strcmp(name, "probe_hbm_read_mbps") || !strcmp(name, "probe_mfma_bf16_gflops") || !strcmp(name, "probe_mfma16_bf16_gflops")) {
    const bool m16 = !strcmp(name, "probe_mfma16_bf16_gflops");
    // measured ceilings of this device; a few milliseconds each, best of 3
    int rc = set_device(h);
    if (rc) return rc;
    if ((rc = ensure(h, h->w_thr, 64))) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(h, hipEventCreate(&e0));
    HIP_TRY(h, hipEventCreate(&e1));
    const bool rd = name[6] == 'h';
    // The READ probes run on the process' default (null) stream: there the bursts read 7.22 TB/s of a 10M-row store, on the
    // handle's own stream 6.98 — while searches run the same 7.16 on either (benchmarks/stream_ab.py).  The difference is
    // not understood; the higher figure is the one no scan has exceeded, so it is the one reported as the measured ceiling.
    hipStream_t ps = rd ? (hipStream_t) nullptr : h->stream;
    const void* src = h->dtype == VDB_BF16 ? (const void*)h->rows_bf16 : (const void*)h->rows_f32;
    const size_t bytes = (size_t)h->n * h->ld * (h->dtype == VDB_BF16 ? 2 : 4);
    if (rd && (!src || bytes < ((size_t)64 << 20))) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail(h, VDB_ERR_ARG, "probe_hbm_read_mbps needs a store of at least 64 MiB to stream"); }
    const int iters = 40000;
    double best = 0.0;
    // read probe: the scan's own launch shape — 12 KiB contiguous per wave and step, non-temporal, 256 threads,
    // 2 or 4 blocks per CU (the scan runs 2) — best of 4 timed launches each after a warm-up launch
    const size_t n16 = bytes / 16 / (12 * 64) * (12 * 64);   // whole 12 KiB tiles
    // The read probes are timed as BURSTS of kBurst launches between two events, like the steps of a search loop: a
    // launch that starts on an idle device (every probe launch did, in rounds 1-2: event, launch, event, synchronise)
    // streams ~2 % under the rate back-to-back launches settle at, and so measured below the very scans it is a ceiling for.
    // And the device must be in its steady state first: after the idle gaps a host leaves between its calls the first tens
    // of milliseconds of launches run ~2 % slow (bench.py settles 150 steps before it times for the same reason), so an
    // untimed run of ~100 ms of reads goes first and each timed burst is ~50 ms long.
    constexpr int kBurst = 24;
    if (rd) {   // ~0.7 s of reads first (benchmarks/stream_ab.py: a 1.25M-row step runs 0.306 ms for the first ~0.4 s of a loop and 0.295 ms from then on)
      const auto t_warm = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_warm).count() < 0.7) {
        for (int i = 0; i < kBurst; ++i) hipLaunchKernelGGL((probe_read_kernel<12>), dim3(h->n_cu * 2), dim3(256), 0, ps, (const v4u_t*)src, n16, (unsigned*)h->w_thr.p);
        HIP_TRY(h, hipStreamSynchronize(ps));
      }
    }
    for (int rep = 0; rep < (rd ? 3 : 4); ++rep) {
      HIP_TRY(h, hipEventRecord(e0, ps));
      const int gm = (rep & 1) ? 2 : 4;
      if (rd) for (int i = 0; i < kBurst; ++i) hipLaunchKernelGGL((probe_read_kernel<12>), dim3(h->n_cu * gm), dim3(256), 0, ps, (const v4u_t*)src, n16, (unsigned*)h->w_thr.p);
      else if (m16) hipLaunchKernelGGL(probe_mfma16_kernel, dim3(h->n_cu), dim3(512), 0, ps, iters, (float*)h->w_thr.p);
      else hipLaunchKernelGGL(probe_mfma_kernel, dim3(h->n_cu), dim3(512), 0, ps, iters, (float*)h->w_thr.p);
      HIP_TRY(h, hipGetLastError());
      HIP_TRY(h, hipEventRecord(e1, ps));
      HIP_TRY(h, hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
      if ((!rd && rep == 0) || ms <= 0.f) continue;   // first launch warms up
      const double v = rd ? (double)kBurst * n16 * 16 / (ms * 1e-3) / 1e6 : (double)h->n_cu * 8 * iters * (m16 ? 16 * 16384.0 : 8 * 32768.0) / (ms * 1e-3) / 1e9;
      best = std::max(best, v);
    }
    if (rd && h->scan_balance == 1 && h->w_ticket.p) {
      // ... and in the fused scan's partition: the tile ranges its self-tuned per-XCD shares deal out (scan_fused.h; the
      // same formula on the same share table), so the figure stays a ceiling for a scan that balances its XCDs
      const int n_blocks = gemv_grid(h);
      const int n_cu = std::max(8, h->n_cu & ~7);
      const int rounds = std::min(kFusedMaxRounds, (n_blocks + n_cu - 1) / n_cu), n_groups = 8 * rounds;
      unsigned share[8 * kFusedMaxRounds] = {};
      HIP_TRY(h, hipDeviceSynchronize());
      HIP_TRY(h, hipMemcpy(share, (unsigned*)h->w_ticket.p + kFusedShareOffset, sizeof share, hipMemcpyDeviceToHost));
      auto group_blocks = [&](int g) {
        const int r = g >> 3, lo = r * n_cu, hi = r == rounds - 1 ? n_blocks : std::min(n_blocks, lo + n_cu);
        return std::max(0, (hi - lo - (g & 7) + 7) >> 3);
      };
      const long long n_tiles = (long long)(n16 / (12 * 64));
      std::vector<long long> before(n_groups + 1, 0);
      for (int g = 0; g < n_groups; ++g) before[g + 1] = before[g] + (long long)(share[g] ? share[g] : kFusedShareUnit) * group_blocks(g);
      std::vector<ProbeRange> rg((size_t)n_blocks);
      const long long R = (long long)std::max(1, h->scan_super_rounds) * n_blocks * 4, full = h->scan_super_rounds > 0 ? n_tiles / R : 0, rem = n_tiles - full * R, total = before[n_groups];
      for (int b = 0; b < n_blocks; ++b) {   // the partition of scan_topk_fused_kernel (scan_fused.h), block by block
        const int round = std::min(b / n_cu, rounds - 1), grp = (b & 7) + 8 * round;
        const long long s_lo = R * before[grp] / total, s_hi = R * before[grp + 1] / total;
        const long long t_lo = rem * before[grp] / total, t_hi = rem * before[grp + 1] / total;
        ProbeRange r;
        r.wid0 = ((b - round * n_cu) >> 3) * 4; r.nwv = group_blocks(grp) * 4;
        r.base = (int)s_lo; r.q = (int)(s_hi - s_lo); r.R = (int)R; r.full_q = (int)(full * (s_hi - s_lo));
        r.tail_base = (int)(full * R + t_lo); r.n_seq = r.full_q + (int)(t_hi - t_lo);
        if (r.q <= 0) { r.q = 1; r.full_q = 0; r.n_seq = (int)(t_hi - t_lo); }
        rg[(size_t)b] = r;
      }
      if ((rc = ensure(h, h->w_idx, rg.size() * sizeof(ProbeRange)))) return rc;
      HIP_TRY(h, hipMemcpy(h->w_idx.p, rg.data(), rg.size() * sizeof(ProbeRange), hipMemcpyHostToDevice));
      for (int rep = 0; rep < 4; ++rep) {
        HIP_TRY(h, hipEventRecord(e0, ps));
        const ProbeRange* rgp = (const ProbeRange*)h->w_idx.p;
        for (int i = 0; i < kBurst; ++i)
        switch (rep & 3) {   // pauses of 0 / 128 / 256 / 384 cycles between a wave's tiles (probes.h)
          case 0: hipLaunchKernelGGL((probe_read_ranges_kernel<12, 0>), dim3(n_blocks), dim3(256), 0, ps, (const v4u_t*)src, rgp, (unsigned*)h->w_thr.p); break;
          case 1: hipLaunchKernelGGL((probe_read_ranges_kernel<12, 2>), dim3(n_blocks), dim3(256), 0, ps, (const v4u_t*)src, rgp, (unsigned*)h->w_thr.p); break;
          case 2: hipLaunchKernelGGL((probe_read_ranges_kernel<12, 4>), dim3(n_blocks), dim3(256), 0, ps, (const v4u_t*)src, rgp, (unsigned*)h->w_thr.p); break;
          default: hipLaunchKernelGGL((probe_read_ranges_kernel<12, 6>), dim3(n_blocks), dim3(256), 0, ps, (const v4u_t*)src, rgp, (unsigned*)h->w_thr.p); break;
        }
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(e1, ps));
        HIP_TRY(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
        if (ms <= 0.f) continue;
        best = std::max(best, (double)kBurst * n_tiles * 12 * 1024 / (ms * 1e-3) / 1e6);
      }
    }
    if (rd && fast_dim(h) && h->fuse_small && gemv_grid(h) <= kFusedMaxParts) {
      // ... and the scan's OWN loop with the selection taken out (lists that accept nothing): the same loads in the same
      // rhythm, the same arithmetic, strictly less work — the bare reads above, which re-issue the moment their data is
      // back, stream 2-4 % below the real scan on this chip, so they are a reference point and THIS is the ceiling
      const int grid = gemv_grid(h);
      const WsSet W = ws_of(h, 0);
      if ((rc = ensure_parts(h, W, (size_t)grid * 64))) return rc;
      if ((rc = ensure(h, h->w_q, (size_t)h->ld * 4))) return rc;
      if ((rc = ensure(h, h->w_out_i, 64 * 12))) return rc;
      HIP_TRY(h, hipMemsetAsync(h->w_q.p, 0, (size_t)h->ld * 4, ps));
      const WsPtrs P = ptrs_of(W);
      for (int rep = 0; rep < 2; ++rep) {
        FusedScanArgs a;
        memset(&a, 0, sizeof a);
        a.rows = h->dtype == VDB_BF16 ? (const void*)h->rows_bf16 : (const void*)h->rows_f32; a.n_rows = (int)h->n;
        a.q = (const float*)h->w_q.p; a.normalize = 0; a.kk = 40; a.k = 10; a.kks = 40; a.self0 = -1;
        a.part_s = P.ps; a.part_i = P.pi; a.ticket = (unsigned*)h->w_ticket.p; a.balance = h->scan_balance; a.super_rounds = h->scan_super_rounds; a.n_cu = h->n_cu;
        a.rerank = 0; a.rows_b16 = h->rows_bf16; a.out_idx = (long long*)h->w_out_i.p; a.out_dist = (float*)((char*)h->w_out_i.p + 64 * 8);
        a.probe = 1;
        HIP_TRY(h, hipEventRecord(e0, ps));
        bool launched = true;
        for (int i = 0; i < kBurst && launched; ++i) launched = launch_scan_fused(h->dtype == VDB_BF16, h->dim, a, 1, h->scan_nt_loads != 0, grid, ps);
        if (!launched) break;
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(e1, ps));
        HIP_TRY(h, hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
        if (ms <= 0.f) continue;
        best = std::max(best, (double)kBurst * bytes / (ms * 1e-3) / 1e6);
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *value = (int64_t)best;
    return VDB_OK;
  }
  if (!strcmp(name, "own_stream_handle")) { *value = (int64_t)(intptr_t)h->stream; return VDB_OK; }   // diagnostic: the handle's own stream, as a caller may pass it back
  if (!strcmp(name, "last_fused")) { *value = h->last_fused; return VDB_OK; }
  if (!strcmp(name, "mfma_last_skipped_tiles")) { *value = h->mfma_last_skipped; return VDB_OK; }
  if (!strncmp(name, "scan_share_", 11)) {   // the fused scan's self-tuned share of block group g (1024 = equal)
    const long g = strtol(name + 11, nullptr, 10);
    if (g < 0 || g >= 8 * kFusedMaxRounds) return fail(h, VDB_ERR_ARG, "scan_share_<g>: g in [0, 32)");
    int rc = set_device(h);
    if (rc) return rc;
    unsigned v = 0;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(&v, (unsigned*)h->w_ticket.p + kFusedShareOffset + g, 4, hipMemcpyDeviceToHost));
    *value = v ? v : kFusedShareUnit;
    return VDB_OK;
  }
  if (!strncmp(name, "scan_stamp_", 11)) {
    const long i = strtol(name + 11, nullptr, 10);
    if (!h->w_stamps.p || i < 0 || i >= 16 + 6 * kFusedMaxParts) return fail(h, VDB_ERR_ARG, "scan_stamp_<i>: set option scan_stamps first; i in [0, 6160)");
    int rc = set_device(h);
    if (rc) return rc;
    static thread_local std::vector<long long> cache;   // one copy serves a run of reads: index 0 refreshes it
    if (i == 0 || cache.empty()) {
      cache.resize(16 + 6 * kFusedMaxParts);
      HIP_TRY(h, hipDeviceSynchronize());
      HIP_TRY(h, hipMemcpy(cache.data(), h->w_stamps.p, cache.size() * 8, hipMemcpyDeviceToHost));
    }
    *value = cache[(size_t)i];
    return VDB_OK;
  }
  if (!strcmp(name, "mfma_last_hits")) {   // candidates appended by the last (full) pass, summed over queries
    *value = 0;
    if (!h->w_cand_cnt.p || h->mfma_last_b <= 0) return VDB_OK;
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<int> cnt((size_t)h->mfma_last_b * kMfCntStrideC);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(cnt.data(), h->w_cand_cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    int64_t tot = 0;
    for (size_t q = 0; q < cnt.size(); q += kMfCntStrideC) tot += cnt[q];
    *value = tot;
    return VDB_OK;
  }
  return fail(h, VDB_ERR_ARG, std::string("get_stat: unknown stat ") + name);
}

int vdb_synth_rows_dev(vdb_t* h, float* out_dev, int64_t row0, int64_t n, const vdb_synth_spec* spec, void* stream) {
  if (!h) return VDB_ERR_ARG;
  if (!spec || n < 0 || (n > 0 && !out_dev)) return fail(h, VDB_ERR_ARG, "synth: bad arguments");
  if (n == 0) return VDB_OK;
  if (is_sharded(h)) return vdb_synth_rows_dev(h->shards[0], out_dev, row0, n, spec, stream);   // shard 0's device
  int rc = set_device(h);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;  // NULL = the default (null) stream, HIP convention
  const int blocks = (int)std::min<int64_t>((n + 3) / 4, (int64_t)h->n_cu * 8);
  hipLaunchKernelGGL(synth_rows_kernel, dim3(blocks), dim3(256), 0, st, out_dev, (long long)row0, (long long)n, h->dim, h->dim, *spec);
  HIP_TRY(h, hipGetLastError());
  return VDB_OK;
}

int vdb_add_synth(vdb_t* h, int64_t row0, int64_t n, const vdb_synth_spec* spec) {
  if (!h) return VDB_ERR_ARG;
  if (!spec || n < 0) return fail(h, VDB_ERR_ARG, "add_synth: bad arguments");
  if (n == 0) return VDB_OK;
  if (is_sharded(h)) {   // every shard generates its own rows in its own HBM (all GPUs, one after the other)
    if (!h->per) return fail(h, VDB_ERR_STATE, "vdb_add_synth: call vdb_reserve(total rows) on a sharded handle first (it fixes the row partition)");
    for (int64_t r = 0; r < n;) {
      const ShardSpan sp = shard_span(h, h->n, n - r);
      vdb_t* c = h->shards[(size_t)sp.s];
      const int rc = vdb_add_synth(c, row0 + r, sp.count, spec);
      if (rc) return fail(h, rc, c->err);
      r += sp.count; h->n += sp.count;
    }
    return VDB_OK;
  }
  int rc = set_device(h);
  if (rc) return rc;
  if ((rc = reserve_for_add(h, n))) return rc;
  const int64_t chunk = std::max<int64_t>(1, (int64_t)(256u << 20) / ((int64_t)h->dim * 4));
  if ((rc = ensure(h, h->w_stage, (size_t)std::min(n, chunk) * h->dim * 4))) return rc;
  for (int64_t r = 0; r < n; r += chunk) {
    const int64_t m = std::min(chunk, n - r);
    const int blocks = (int)std::min<int64_t>((m + 3) / 4, (int64_t)h->n_cu * 8);
    hipLaunchKernelGGL(synth_rows_kernel, dim3(blocks), dim3(256), 0, h->stream, (float*)h->w_stage.p, (long long)(row0 + r), (long long)m, h->dim, h->dim, *spec);
    HIP_TRY(h, hipGetLastError());
    if ((rc = normalize_into_store(h, (const float*)h->w_stage.p, h->dim, m, h->stream))) return rc;
    h->n += m;
  }
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return VDB_OK;
}

}  // extern "C"
