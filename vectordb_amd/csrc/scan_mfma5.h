// scan_mfma5.h — the wide tile of the batched pass kernel (256 corpus rows x 256 queries per block): its rationale, LDS
// layout and constants.  The kernel itself is scan_mfma6.h (v_mfma_f32_16x16x32_bf16); the first form of this tile, on
// 32x32x16 MFMAs with the same ring, barriers and epilogue, ran 1.3-1.9 % slower (the chip holds a lower clock on that
// shape) and was retired in round 2 — what is said below about sub-steps of k16 describes it, scan_mfma6.h restates the
// K-step for 16x16x32.
//
// Why: 256x128 tiles (the round-1 kernels; today scan_mfma6.h with NB = 2, for batches of at most 128 queries) stage (1/128 + 1/256) bytes per
// multiply-add column pair through LDS — 184 GB for 10M x 768 against 1024 queries — and the
// L2 -> LDS path delivers at most ~56 GB/s per CU (14.3 TB/s chip-wide; benchmarks/
// lds_dma_probe.hip): 12.9 ms of operand delivery, longer than the MFMA work itself.  A
// 256x256 tile stages 123 GB (8.6 ms) and halves the LDS fragment reads per MFMA (a wave owns
// 128 x 64: 6 ds_read_b128 feed 8 MFMAs instead of 4 feeding 4).
//
// The same probe sets the shape of the staging ring.  That rate needs loads in flight ACROSS the
// block barrier: a two-buffer kernel that refills right behind its barrier and drains before the
// next one (64 KiB burst, then nothing) gets 39.8 GB/s per CU; with 32 KiB still in flight at
// every barrier it is 56.  Half-line (64 B per row) pieces, i.e. K = 32 stages, get 37: the
// stage stays K = 64.  Three whole 64 KiB stages do not fit in 160 KiB of LDS, so the operands
// ring separately: THREE 32 KiB corpus slots and TWO 32 KiB query slots (160 KiB, all of it).
//
// Structure (512 threads = 8 waves, 2(M) x 4(N), two waves per SIMD, <= 256 registers):
//   * step t multiplies corpus slot t%3 by query slot t%2; the 12 K-steps of a tile are
//     unrolled (slots, the first step's C = 0 instead of a 128-register clear: compile-time);
//   * ONE block barrier per K-step, before the last of its four k16 sub-steps.  Each wave first
//     waits until only its newest 4 loads (corpus stage t+2, issued in this step) are
//     outstanding, so the barrier certifies that stage t+1 has landed and that everybody's reads
//     of stage t have returned; behind it the wave refills the query slot just released (stage
//     t+2); the corpus slot released there is refilled (stage t+3) in the next step's first
//     sub-step.  Issue order B(t+1), A(t+2), B(t+2), A(t+3) ... makes the counted wait exact;
//   * fragments are double-buffered in registers: the reads of sub-step j+1 (across the stage
//     boundary too) are issued between the MFMAs of sub-step j, so no MFMA waits on LDS;
//   * epilogue: VALU time is what it costs (two waves share a SIMD and reach it together), so it
//     is a max tree: per 32x32 block the lane's 16 scores reduce to four group maxima and one
//     block maximum; one ballot per block, then one per hot group of four, and only those four
//     scores are examined (the trees run on the scores' bit patterns, see thr_bits below).  A hit
//     waits in its lane's registers (two places) until some lane needs a third; then all waiting
//     hits reserve their slots at once (one atomic round trip per flush, not per hit).
//
// Where the time goes (10M x 768, B = 1024, final pass; rocprofv3 PMC, cycles per K-step and CU):
// MFMA alone 2100 (97 % of the matrix pipe, 2.0 GHz), staging alone 2580 (25 B/clk/CU through the
// texture path: the binding rate is in shader clocks), both 2920 at ~1.7 GHz.  The kernel is
// bound by operand delivery into LDS, 13 % above that floor.
// Thresholds, append buffers, exactness argument and fallback are those of scan_mfma3.h.
#pragma once
#include <type_traits>
#include "scan_mfma3.h"

namespace vdbk {

constexpr int kM5BN = 256;                          // queries per block
constexpr int kM5Slot = kMfBM * kMfBK * 2;          // 32 KiB: 256 rows x 128 B, either operand
constexpr int kM5SlotsA = 3;
constexpr int kM5SlotsB = 2;
constexpr int kM5BaseB = kM5SlotsA * kM5Slot;
constexpr int kM5Threads = 512;
static_assert(kM5BN == kMfBM, "one set of staging offsets serves both operands");

__host__ __device__ constexpr size_t m5_smem_bytes() { return (size_t)(kM5SlotsA + kM5SlotsB) * kM5Slot; }   // 160 KiB

}  // namespace vdbk
