// lds_dma_probe.hip — what the L2 -> LDS path (global_load_lds_dwordx4) delivers per CU as a
// function of (a) bytes kept in flight and (b) the row-segment width of a piece.  Design input
// for the staging of scan_mfma5.h; not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o lds_dma_probe lds_dma_probe.hip && ./lds_dma_probe
//
// One 512-thread block per CU.  Every wave loops over "stages"; a stage is SLOTS 1 KiB pieces
// per wave.  KEEP stages stay in flight across each step's barrier (counted s_waitcnt vmcnt), so bytes in flight per CU
// = 8 waves x RING x SLOTS KiB.  SEG = 128: a piece covers 8 rows x 128 B (one full line per row
// and K-step).  SEG = 64: a piece covers 16 rows x 64 B and the other half of each line is
// fetched by the NEXT stage (what a K=32 stage would do).  Source: a [rows][1536 B] matrix; all
// blocks of an XCD walk the same rows (L2-shared, as the kernel's query/corpus tiles are).
// PRIVATE = 1: every block streams its own L2-resident 96 KiB instead — no line is requested by
// two CUs.  It reads at the same ~57-59 GB/s per CU, so the rate is the L2 -> L1 fill path of a
// CU, not contention between CUs for shared lines.  (A footprint that fits the 32 KiB L1 reads
// at ~150 GB/s per CU; nothing on the kernel's staging path is that small.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_blocks = 256;   // argv[1]: fewer blocks = fewer CUs pulling from each XCD's L2
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// AUX = cache-policy bits of the load: 1 = sc0, 2 = nt, 16 = sc1
template <int AUX = 0>
__device__ __forceinline__ void glds16(const unsigned char* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

template <int SEG, int KEEP, int SLOTS, int PRIVATE = 0, int AUX = 0>
__global__ __launch_bounds__(512) void probe(const unsigned char* __restrict__ src, long long n_rows, int n_stages, int rows_per_block_stage) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int RING = KEEP + 1;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  // rows walked by this XCD's blocks: a window that advances by rows_per_block_stage per stage
  const long long xcd_rows = n_rows / 8;
  const unsigned char* base = src + (size_t)xcd * xcd_rows * 1536;
  constexpr int ROWS_PER_PIECE = 1024 / SEG;
  const unsigned lane_off = (unsigned)((lane / (SEG / 16)) * 1536 + (lane % (SEG / 16)) * 16);
  auto issue = [&](int stage) {
    // SEG=128: K-step = stage % 12, row window advances every 12 stages.  SEG=64: 24 half steps.
    constexpr int KSTEPS = 1536 / SEG;
    const int ks = stage % KSTEPS;
    unsigned char* st = smem + ((stage % RING) * 8 + w) * SLOTS * 1024;
    if (PRIVATE) {
      // every block streams its OWN 96 KiB region round and round, as dense 128-byte lines: L2-resident (3 MiB per
      // XCD), three times the 32 KiB L1, and no line is ever requested by two CUs
      const unsigned char* reg = src + (size_t)blockIdx.x * (96 << 10);
      const size_t off = ((size_t)stage * 8 * SLOTS * 1024) % (96 << 10);
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) glds16<AUX>(reg + (off + (size_t)(w * SLOTS + i) * 1024) % (96 << 10) + lane * 16, st + i * 1024);
      return;
    }
    const long long win = ((long long)(stage / KSTEPS) * rows_per_block_stage) % (xcd_rows - rows_per_block_stage);
    const unsigned char* p = base + (size_t)win * 1536 + (size_t)ks * SEG + (size_t)w * SLOTS * ROWS_PER_PIECE * 1536;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) glds16<AUX>(p + (size_t)i * ROWS_PER_PIECE * 1536 + lane_off, st + i * 1024);
  };
#pragma unroll
  for (int s = 0; s < KEEP; ++s) issue(s);
  for (int s = 0; s < n_stages; ++s) {
    issue(s + KEEP);
    // stage s has landed once only the KEEP younger stages are outstanding (KEEP = 0: the
    // burst-then-drain pattern of a two-buffer kernel that refills right behind its barrier)
    if constexpr (KEEP * SLOTS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (KEEP * SLOTS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (KEEP * SLOTS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (KEEP * SLOTS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (KEEP * SLOTS == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Register staging of the same 128-B-segment pattern: global_load_dwordx4 into VGPRs, one stage (SLOTS KiB per
// wave) kept in flight, then ds_write_b128 into LDS (TO_LDS = 1) or folded into a checksum (TO_LDS = 0).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int SLOTS, int TO_LDS>
__global__ __launch_bounds__(512) void probe_regs(const unsigned char* __restrict__ src, long long n_rows, int n_stages,
                                                  int rows_per_block_stage, unsigned* __restrict__ sink) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const long long xcd_rows = n_rows / 8;
  const unsigned char* base = src + (size_t)xcd * xcd_rows * 1536;
  const unsigned lane_off = (unsigned)((lane / 8) * 1536 + (lane % 8) * 16);
  u32x4_t buf[2][SLOTS];
  u32x4_t acc = {0u, 0u, 0u, 0u};
  auto issue = [&](int stage, u32x4_t* dst) {
    const int ks = stage % 12;
    const long long win = ((long long)(stage / 12) * rows_per_block_stage) % (xcd_rows - rows_per_block_stage);
    const unsigned char* p = base + (size_t)win * 1536 + (size_t)ks * 128 + (size_t)w * SLOTS * 8 * 1536;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
      dst[i] = *reinterpret_cast<const u32x4_t*>(p + (size_t)i * 8 * 1536 + lane_off);
  };
  auto consume = [&](int stage, u32x4_t* v) {
    if (TO_LDS) {
      unsigned char* st = smem + ((stage & 1) * 8 + w) * SLOTS * 1024;
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) *reinterpret_cast<u32x4_t*>(st + i * 1024 + lane * 16) = v[i];
    } else {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) acc ^= v[i];
    }
  };
  issue(0, buf[0]);
  for (int s = 0; s < n_stages; s += 2) {
    issue(s + 1, buf[1]);
    consume(s, buf[0]);          // the compiler waits for exactly the older stage's loads (vmcnt = SLOTS)
    __builtin_amdgcn_s_barrier();
    issue(s + 2, buf[0]);
    consume(s + 1, buf[1]);
    __builtin_amdgcn_s_barrier();
  }
  consume(0, buf[0]);
  if (TO_LDS) { __syncthreads(); acc.x = *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4); }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;
}

template <int SLOTS, int TO_LDS>
void run_regs(const unsigned char* d, long long n_rows, unsigned* sink, const char* what) {
  const int n_stages = 6000;
  const int rows_per_block_stage = 8 * SLOTS * 8;
  const size_t smem = (size_t)2 * 8 * SLOTS * 1024;
  CHECK(hipFuncSetAttribute((const void*)probe_regs<SLOTS, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_regs<SLOTS, TO_LDS>), dim3(g_blocks), dim3(512), smem, 0, d, n_rows, n_stages, rows_per_block_stage, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)g_blocks * n_stages * 8 * SLOTS * 1024;
  printf("%-34s seg 128 B  kept in flight/CU %3d KiB  stage %2d KiB : %6.2f ms  %6.2f TB/s  %5.1f GB/s per CU\n", what,
         8 * SLOTS, 8 * SLOTS, ms, bytes / ms / 1e9, bytes / ms / 1e6 / g_blocks);
}

// The SEG = 128 pattern again, issued as buffer_load_dwordx4 ... lds (SGPR resource + 32-bit per-lane offset + scalar
// offset) instead of global_load_lds_dwordx4 (64-bit per-lane address): does the address form change the fill rate?
template <int KEEP, int SLOTS>
__global__ __launch_bounds__(512) void probe_buf(const unsigned char* __restrict__ src, long long n_rows, int n_stages, int rows_per_block_stage) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int RING = KEEP + 1;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const long long xcd_rows = n_rows / 8;
  const unsigned char* base = src + (size_t)xcd * xcd_rows * 1536;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(xcd_rows * 1536), 0x00020000);
  const unsigned lane_off = (unsigned)((lane / 8) * 1536 + (lane % 8) * 16);
  auto issue = [&](int stage) {
    const int ks = stage % 12;
    unsigned char* st = smem + ((stage % RING) * 8 + w) * SLOTS * 1024;
    const long long win = ((long long)(stage / 12) * rows_per_block_stage) % (xcd_rows - rows_per_block_stage);
    const unsigned soff = (unsigned)((size_t)win * 1536 + (size_t)ks * 128 + (size_t)w * SLOTS * 8 * 1536);
#pragma unroll
    for (int i = 0; i < SLOTS; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(st + i * 1024), 16, lane_off,
                                           soff + (unsigned)(i * 8 * 1536), 0, 0);
  };
#pragma unroll
  for (int s = 0; s < KEEP; ++s) issue(s);
  for (int s = 0; s < n_stages; ++s) {
    issue(s + KEEP);
    if constexpr (KEEP * SLOTS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (KEEP * SLOTS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int KEEP, int SLOTS>
void run_buf(const unsigned char* d, long long n_rows, const char* what) {
  const int n_stages = 6000;
  const int rows_per_block_stage = 8 * SLOTS * 8;
  const size_t smem = (size_t)(KEEP + 1) * 8 * SLOTS * 1024;
  CHECK(hipFuncSetAttribute((const void*)probe_buf<KEEP, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe_buf<KEEP, SLOTS>), dim3(g_blocks), dim3(512), smem, 0, d, n_rows, n_stages, rows_per_block_stage);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)g_blocks * n_stages * 8 * SLOTS * 1024;
  printf("%-34s seg 128 B  kept in flight/CU %3d KiB  stage %2d KiB : %6.2f ms  %6.2f TB/s  %5.1f GB/s per CU\n", what,
         KEEP * 8 * SLOTS, 8 * SLOTS, ms, bytes / ms / 1e9, bytes / ms / 1e6 / g_blocks);
}

template <int SEG, int KEEP, int SLOTS, int PRIVATE = 0, int AUX = 0>
void run(const unsigned char* d, long long n_rows, const char* what) {
  const int n_stages = 6000;
  const int rows_per_block_stage = 8 * SLOTS * (1024 / SEG);   // rows one block covers per stage
  const size_t smem = (size_t)(KEEP + 1) * 8 * SLOTS * 1024;
  CHECK(hipFuncSetAttribute((const void*)probe<SEG, KEEP, SLOTS, PRIVATE, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<SEG, KEEP, SLOTS, PRIVATE, AUX>), dim3(g_blocks), dim3(512), smem, 0, d, n_rows, n_stages, rows_per_block_stage);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)g_blocks * n_stages * 8 * SLOTS * 1024;
  printf("%-34s seg %3d B  kept in flight/CU %3d KiB  stage %2d KiB : %6.2f ms  %6.2f TB/s  %5.1f GB/s per CU\n", what, SEG,
         KEEP * 8 * SLOTS, 8 * SLOTS, ms, bytes / ms / 1e9, bytes / ms / 1e6 / g_blocks);
}

int main(int argc, char** argv) {
  if (argc > 1) g_blocks = atoi(argv[1]);
  if (g_blocks < 8 || g_blocks > 256) { printf("blocks must be in [8,256]\n"); return 1; }
  printf("%d blocks (one per CU)\n", g_blocks);
  const long long n_rows = 1 << 20;   // 1.5 GiB matrix
  unsigned char* d;
  CHECK(hipMalloc(&d, (size_t)n_rows * 1536));
  CHECK(hipMemset(d, 1, (size_t)n_rows * 1536));
  run<128, 0, 8>(d, n_rows, "64 KiB burst, drained each step");
  run<128, 1, 8>(d, n_rows, "64 KiB stages, 1 kept");
  run<128, 1, 4>(d, n_rows, "32 KiB stages, 1 kept");
  run<128, 2, 4>(d, n_rows, "32 KiB stages, 2 kept");
  run<128, 3, 4>(d, n_rows, "32 KiB stages, 3 kept");
  run<128, 1, 8, 1>(d, n_rows, "64 KiB stages, 1 kept, PRIVATE rows");
  run<128, 2, 4, 1>(d, n_rows, "32 KiB stages, 2 kept, PRIVATE rows");
  run<64, 0, 4>(d, n_rows, "32 KiB half-line burst, drained");
  run<64, 1, 4>(d, n_rows, "32 KiB half-line stages, 1 kept");
  run<64, 2, 4>(d, n_rows, "32 KiB half-line stages, 2 kept");
  run<64, 3, 4>(d, n_rows, "32 KiB half-line stages, 3 kept");
  run_buf<1, 8>(d, n_rows, "buffer_load lds, 64 KiB, 1 kept");
  run_buf<2, 4>(d, n_rows, "buffer_load lds, 32 KiB, 2 kept");
  run<128, 1, 8>(d, n_rows, "64 KiB stages, 1 kept (again)");
  run<128, 1, 8, 0, 1>(d, n_rows, "64 KiB stages, 1 kept, sc0");
  run<128, 1, 8, 0, 2>(d, n_rows, "64 KiB stages, 1 kept, nt");
  run<128, 1, 8, 0, 3>(d, n_rows, "64 KiB stages, 1 kept, sc0 nt");
  run<128, 1, 8, 0, 16>(d, n_rows, "64 KiB stages, 1 kept, sc1");
  run<128, 1, 8, 0, 17>(d, n_rows, "64 KiB stages, 1 kept, sc0 sc1");
  unsigned* sink;
  CHECK(hipMalloc(&sink, 4096));
  run_regs<8, 0>(d, n_rows, sink, "regs only, 64 KiB stages, 1 kept");
  run_regs<8, 1>(d, n_rows, sink, "regs -> ds_write, 64 KiB, 1 kept");
  run_regs<4, 0>(d, n_rows, sink, "regs only, 32 KiB stages, 1 kept");
  run_regs<4, 1>(d, n_rows, sink, "regs -> ds_write, 32 KiB, 1 kept");
  CHECK(hipFree(sink));
  CHECK(hipFree(d));
  return 0;
}
