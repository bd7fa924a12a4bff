// launch.h — host-side seams between the translation units of libvdb_hip.so.
// The kernels are heavy template families (72 GEMV-scan instantiations, the MFMA pass kernels);
// each family compiles in its own TU (k_gemv.hip; k_mfma.hip and, behind mfma_pass.h, k_mfma_wide16.hip, k_stream.hip) so that the library
// builds in parallel and a change to one kernel rebuilds one file.  vdb_api.hip (the C ABI and all
// orchestration) sees only these plain functions.  No device symbol crosses a TU.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vdbk {

// ---- k_gemv.hip: the HBM-bound scan (scan_gemv.h) ---------------------------------------------
struct ScanArgs {
  const void* rows; int n_rows; const float* qn; int kk; int self0; const unsigned* allow; float* part_s; int* part_i;
  int grid; hipStream_t st; size_t smem;
  int xcd_slices = 1;        // fast kernel only: blocks of one XCD take adjacent tiles (0: block-by-block round-robin)
  int ny = 1;                // fast kernel only: groups of qb queries in this launch (grid.y); group g reads qn + g*qb*dim, writes part + g*qb*grid*kk
};
// compile-time row length (768 or 384), qb in {1,2,4} queries per corpus pass, e = list entries per lane
bool launch_scan_fast(bool bf16, int dim, const ScanArgs& a, int qb, int e, bool nt);
// any padded row length ld, one query per pass
bool launch_scan_any(bool bf16, const ScanArgs& a, int ld, int e, bool nt);

// the whole search of <= 4 queries with k' <= 64 in one launch (scan_fused.h): normalise, scan, fold, rerank, rank
struct FusedScanArgs {
  const void* rows; int n_rows;
  const float* q;            // [qb][dim] queries: raw (normalize != 0) or already unit vectors
  int normalize;
  int kk, k, kks;            // k' (list length), k (outputs), list stride in the partial buffers (kk rounded up to 4)
  int self0; const unsigned* allow;
  float* part_s; int* part_i;      // [qb][grid][kks]
  unsigned* ticket;                // kFusedTicketBytes, zero before the first launch: the arrival counter (the last arriver resets it), and
                                   // on lines of their own the per-class scan-time sums and the class shares that tune themselves
  int balance;                     // != 0: deal each group of blocks (same XCD, same turn on its CU) a range of tiles by its self-tuned share
  int super_rounds;                // > 0: the groups' tiles interleave in super-rounds of this many tiles per wave of the grid (default kFusedTilesPerWaveAndRoundC), one window over the corpus; 0: one contiguous range per group (round 3's first form, for A/B)
  int n_cu;                        // compute units of the device (blocks beyond the first n_cu are a CU's second, third, ... block)
  int rerank; const float* rows_f32; const uint16_t* rows_b16;   // fp32 copy (or nullptr: re-score against the bf16 rows)
  long long row_offset;
  long long* out_idx; float* out_dist;   // [qb][k]
  int probe;                       // != 0 (stat probe_hbm_read_mbps): the lists accept nothing — the scan's own loop, loads and arithmetic, with the selection out
  long long* stamps;               // diagnostic (option "scan_stamps"): 100 MHz clock stamps, [16] of the last block + [grid][6]; else nullptr
};
constexpr int kFusedTicketBytes = 512;
constexpr int kFusedMaxRounds = 4;          // block groups: 8 XCD classes x up to 4 rounds of blocks over the CUs
constexpr int kFusedDurOffset = 32;         // unsigneds from the ticket: [32] sums of block scan times (100 MHz ticks) per group
constexpr int kFusedShareOffset = 64;       // [32] group shares (0 = not tuned yet = kFusedShareUnit)
constexpr int kFusedShareUnit = 1024;
constexpr int kFusedTilesPerWaveAndRoundC = 8;   // super-round of the fused scan's tile partition: this many tiles per wave of the grid
constexpr int kFusedMaxParts = 1024;        // lists per query (= scan blocks) the fused epilogue covers
// dim 768 or 384, qb in {1,2,4}, k' <= 64, grid <= kFusedMaxParts; false = no such instantiation
bool launch_scan_fused(bool bf16, int dim, const FusedScanArgs& a, int qb, bool nt, int grid, hipStream_t st);

// ---- k_mfma.hip: the batched bf16 path (scan_mfma{,3,4,5}.h) ----------------------------------
// One batched search over device-resident, normalised fp32 queries: dense sample -> thresholds -> append
// passes -> exact top-k' lists (list_*), with the exact list kernel as the device-flagged overflow fallback
// (part_*).  Every buffer is the caller's; nothing is allocated here.
struct MfSearchArgs {
  const unsigned char* rows_bf16; int n_rows; const float* qn; int b; int kk; int self0;
  int n_cu; int variant;           // 4 = 256x128 warp-specialised, 5 = 256x256 wide tile, 6 = streaming (<= 64 queries)
  int cap; int split_x1000; int skip_sample; int force_fallback; int ablate; int stream_nt;
  int one_pass_span;               // stores up to this many times the dense sample are searched in ONE append pass
  int dense_tiles;                 // 256-row tiles of the dense sample that gives the first threshold (<= 64)
  unsigned char* qb16;             // [b_pad][768] bf16
  int qb16_ready;                  // != 0: the caller filled qb16 (and cleared the overflow flag) already
  float* cand_s; int* cand_i;      // [b][cap]
  int* cand_cnt;                   // [b] counters, kMfCntStrideC ints apart, + the overflow flag behind them
  float* thr;                      // [b]
  float* dense;                    // [b][kMf3DenseLd]
  float* list_s; int* list_i;      // [b][kk]   result lists
  float* seed_s; int* seed_i;      // [b][kk]
  float* part_s; int* part_i;      // [b][fb_splits][kk]   fallback partial lists
  hipStream_t st;
};
struct MfPlan { int variant; int n_qt; int b_pad; int n_qt_fb; int fb_splits; };
constexpr int kMfDenseLdC = 32768;      // floats per query of the dense sample's buffer (up to 128 tiles x 256 rows; k_mfma.hip asserts it)
constexpr int kMfCntStrideC = 32;        // ints between two queries' append counters (one cache line each)
constexpr int kMfRowBytesC = 1536;      // the batched path serves 768-d bf16 rows
constexpr int kMfMaxQueriesPerCall = 4096;   // mfma_search_launch serves at most this many queries (the caller chunks)
MfPlan mfma_plan(int b, int n_rows, int n_cu, int variant_opt);
hipError_t mfma_prepare_device();                    // raise the dynamic-LDS limits of the kernels on the current device
// returns hipSuccess or the launch error; *skipped_tiles = tiles the final pass left out
hipError_t mfma_search_launch(const MfSearchArgs& a, int* skipped_tiles);
bool mfma_kk_supported(int kk);

}  // namespace vdbk
