// placement_probe.hip — does WHERE a large buffer lands in HBM (how the driver backed it: physical chunk size, a
// fragmented heap) change the rate a streaming read gets?  Background: a 12.5M x 768 bf16 store (19.2 GB) keeps its
// scan rate for its lifetime, but stores made one after the other on one box differ by up to 4 %
// (profiles/r03_size_sweep.txt).  Not part of the library.
//
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip && ./placement_probe [GiB]
//
// The read: 512 blocks x 256 threads, each block a contiguous range, 12 non-temporal 16-byte loads in flight per
// thread (the scan's shape).  Each variant: 0.5 s of warm-up, then 5 bursts of 20 launches between events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink) {
  const size_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n_vec ? lo + per : n_vec;
  u32x4 acc = {0u, 0u, 0u, 0u};
  size_t i = lo + threadIdx.x;
  for (; i + 11 * 256 < hi; i += 12 * 256) {
    u32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = __builtin_nontemporal_load(p + i + (size_t)j * 256);
#pragma unroll
    for (int j = 0; j < 12; ++j) acc ^= v[j];
  }
  for (; i < hi; i += 256) acc ^= __builtin_nontemporal_load(p + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1u;   // never true for the fill pattern: keeps the loads
}

// the same bytes as ONE moving window: 48 KiB pieces dealt round-robin to the blocks
__global__ __launch_bounds__(256) void read_rr_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t piece = 12 * 256;
  for (size_t base = (size_t)blockIdx.x * piece; base + piece <= n_vec; base += (size_t)gridDim.x * piece) {
    u32x4 v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = __builtin_nontemporal_load(p + base + threadIdx.x + (size_t)j * 256);
#pragma unroll
    for (int j = 0; j < 12; ++j) acc ^= v[j];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1u;
}
// G moving windows: the buffer in G contiguous ranges, each dealt round-robin (48 KiB pieces) to gridDim.x / G blocks;
// spread != 0: block b belongs to group b % G (the fused scan's classes: XCD = b & 7), else to group b / (gridDim.x / G)
__global__ __launch_bounds__(256) void read_groups_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink, int G, int spread) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t piece = 12 * 256;
  const int per_g = (int)gridDim.x / G;
  const int g = spread ? (int)blockIdx.x % G : (int)blockIdx.x / per_g, j = spread ? (int)blockIdx.x / G : (int)blockIdx.x % per_g;
  const size_t pieces = n_vec / piece, lo = pieces * g / G, hi = pieces * (g + 1) / G;
  for (size_t t = lo + j; t < hi; t += per_g) {
    const size_t base = t * piece;
    u32x4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = __builtin_nontemporal_load(p + base + threadIdx.x + (size_t)k * 256);
#pragma unroll
    for (int k = 0; k < 12; ++k) acc ^= v[k];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1u;
}
// ONE moving window made of per-XCD slices: super-rounds of 8 x 64 x c pieces; in each, XCD class g = b & 7 owns the
// contiguous slice [g * 64c, (g + 1) * 64c) and deals it round-robin to its 64 blocks (c pieces each)
__global__ __launch_bounds__(256) void read_super_kernel(const u32x4* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink, int c) {
  u32x4 acc = {0u, 0u, 0u, 0u};
  const size_t piece = 12 * 256, pieces = n_vec / piece;
  const int g = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, per_g = (int)gridDim.x >> 3;
  const size_t R = (size_t)8 * per_g * c;
  for (size_t r = 0; r * R < pieces; ++r)
    for (int k = 0; k < c; ++k) {
      const size_t t = r * R + (size_t)g * per_g * c + (size_t)k * per_g + j;
      if (t >= pieces) break;
      const size_t base = t * piece;
      u32x4 v[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) v[q] = __builtin_nontemporal_load(p + base + threadIdx.x + (size_t)q * 256);
#pragma unroll
      for (int q = 0; q < 12; ++q) acc ^= v[q];
    }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1u;
}
static int g_rr = 0, g_groups = 0, g_spread = 0, g_super = 0;

static double time_reads(const void* buf, size_t bytes, unsigned* sink) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t n_vec = bytes / 16;
  const int warm = (int)std::max<size_t>(8, (size_t)(0.5 * 7e12 / (double)bytes));
  auto launch = [&]() {
    if (g_super) hipLaunchKernelGGL(read_super_kernel, dim3(512), dim3(256), 0, 0, (const u32x4*)buf, n_vec, sink, g_super);
    else if (g_groups) hipLaunchKernelGGL(read_groups_kernel, dim3(512), dim3(256), 0, 0, (const u32x4*)buf, n_vec, sink, g_groups, g_spread);
    else if (g_rr) hipLaunchKernelGGL(read_rr_kernel, dim3(512), dim3(256), 0, 0, (const u32x4*)buf, n_vec, sink);
    else hipLaunchKernelGGL(read_kernel, dim3(512), dim3(256), 0, 0, (const u32x4*)buf, n_vec, sink);
  };
  for (int i = 0; i < warm; ++i) launch();
  CHECK(hipDeviceSynchronize());
  std::vector<double> r;
  for (int b = 0; b < 5; ++b) {
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    r.push_back((double)bytes * 20 / (ms * 1e-3) / 1e9);
  }
  std::sort(r.begin(), r.end());
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return r[2];
}

struct VmmBuf { void* va = nullptr; size_t bytes = 0; std::vector<hipMemGenericAllocationHandle_t> hs; };

static bool vmm_alloc(VmmBuf& b, size_t bytes, size_t chunk, int dev) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess) { (void)hipGetLastError(); return false; }
  chunk = (chunk + gran - 1) / gran * gran;
  bytes = (bytes + chunk - 1) / chunk * chunk;
  if (hipMemAddressReserve(&b.va, bytes, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  b.bytes = bytes;
  for (size_t off = 0; off < bytes; off += chunk) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu MiB\n", off >> 20); (void)hipGetLastError(); return false; }
    b.hs.push_back(h);
    if (hipMemMap((char*)b.va + off, chunk, 0, h, 0) != hipSuccess) { printf("hipMemMap failed\n"); (void)hipGetLastError(); return false; }
  }
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(b.va, bytes, &acc, 1) != hipSuccess) { printf("hipMemSetAccess failed\n"); (void)hipGetLastError(); return false; }
  return true;
}
static void vmm_free(VmmBuf& b) {
  if (b.va) { (void)hipMemUnmap(b.va, b.bytes); }
  for (auto h : b.hs) (void)hipMemRelease(h);
  if (b.va) (void)hipMemAddressFree(b.va, b.bytes);
  b = VmmBuf();
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 17.88;   // 19.2e9 bytes
  const size_t bytes = (size_t)(gib * (double)(1ull << 30)) / (2u << 20) * (2u << 20);
  CHECK(hipSetDevice(0));
  unsigned* sink; CHECK(hipMalloc(&sink, 256)); CHECK(hipMemset(sink, 0, 256));
  printf("buffer %.2f GB\n", bytes / 1e9);
  if (argc > 2) {   // windows of ONE large allocation on a fresh heap: is the rate a property of the region?
    const int n_win = atoi(argv[2]);
    char* big; CHECK(hipMalloc((void**)&big, bytes * (size_t)n_win)); CHECK(hipMemset(big, 0x5a, bytes * (size_t)n_win));
    for (int rep = 0; rep < 1; ++rep)
      for (int k = 0; k < n_win; ++k) {
        g_rr = 0; const double a = time_reads(big + (size_t)k * bytes, bytes, sink);
        g_rr = 1; const double b = time_reads(big + (size_t)k * bytes, bytes, sink);
        g_rr = 0;
        printf("window %2d of one %.0f GB allocation (%d)   : %7.1f GB/s as 512 ranges, %7.1f as one moving window\n", k, n_win * bytes / 1e9, rep, a, b);
      }
    // how many concurrent windows does it take?  slow and fast windows of the list above, G = 1 ... 512 groups
    for (int k : {0, 3, 7, 9, 10}) {
      if (k >= n_win) continue;
      printf("window %2d, G moving windows (blocks b/(512/G) | b%%G):", k);
      for (int G : {1, 2, 4, 8, 16, 32, 64, 128, 512}) {
        g_groups = G; g_spread = 0; const double a = time_reads(big + (size_t)k * bytes, bytes, sink);
        g_spread = 1; const double b = time_reads(big + (size_t)k * bytes, bytes, sink);
        printf("  G=%d %6.0f|%6.0f", G, a, b);
      }
      g_groups = 0; g_spread = 0;
      printf("\n");
      printf("window %2d, one window of per-XCD slices, c pieces per block and super-round:", k);
      for (int c : {1, 2, 4, 8, 16, 64}) { g_super = c; printf("  c=%d %6.0f", c, time_reads(big + (size_t)k * bytes, bytes, sink)); }
      g_super = 0;
      printf("\n");
    }
    CHECK(hipFree(big));
    CHECK(hipFree(sink));
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes));
    printf("hipMalloc, fresh heap (%d)               : %7.1f GB/s\n", rep, time_reads(p, bytes, sink));
    CHECK(hipFree(p));
  }
  const size_t chunks[] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30};
  for (size_t c : chunks) {
    VmmBuf b;
    if (!vmm_alloc(b, bytes, c, 0)) { printf("VMM chunks of %zu MiB: not available\n", c >> 20); vmm_free(b); continue; }
    CHECK(hipMemset(b.va, 0x5a, b.bytes));
    printf("VMM, physical chunks of %5zu MiB          : %7.1f GB/s\n", c >> 20, time_reads(b.va, bytes, sink));
    vmm_free(b);
  }
  // behind a neighbour of growing size: does the rate follow the buffer's offset in the heap?
  {
    const double pads_gb[] = {0.0, 19.2, 38.4, 76.8, 153.6, 230.4, 38.4, 0.0};
    for (double pg : pads_gb) {
      void* pad = nullptr;
      const size_t pb = (size_t)(pg * 1e9) / (2u << 20) * (2u << 20);
      if (pb && hipMalloc(&pad, pb) != hipSuccess) { (void)hipGetLastError(); printf("pad %.1f GB: no memory\n", pg); continue; }
      void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes));
      printf("hipMalloc behind a %5.1f GB neighbour        : %7.1f GB/s\n", pg, time_reads(p, bytes, sink));
      CHECK(hipFree(p));
      if (pad) CHECK(hipFree(pad));
    }
  }
  // a fragmented heap: 3000 buffers of 24 MiB, every other one freed, then the big allocation (the holes are 24 MiB)
  {
    std::vector<void*> small(3000, nullptr);
    for (auto& s : small) CHECK(hipMalloc(&s, (size_t)24 << 20));
    for (size_t i = 0; i < small.size(); i += 2) { CHECK(hipFree(small[i])); small[i] = nullptr; }
    void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes));
    printf("hipMalloc behind 1500 x 24 MiB holes       : %7.1f GB/s\n", time_reads(p, bytes, sink));
    CHECK(hipFree(p));
    for (auto s : small) if (s) CHECK(hipFree(s));
  }
  // store after store, as a process that rebuilds its index does: 57.6 GB (the store's fp32 copy + bf16 rows) allocated and freed between
  for (int rep = 0; rep < 4; ++rep) {
    void* big; CHECK(hipMalloc(&big, bytes * 2));
    void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes));
    printf("hipMalloc beside a %.0f GB neighbour (%d)    : %7.1f GB/s\n", 2 * bytes / 1e9, rep, time_reads(p, bytes, sink));
    CHECK(hipFree(p)); CHECK(hipFree(big));
  }
  for (int rep = 0; rep < 2; ++rep) {
    void* p; CHECK(hipMalloc(&p, bytes)); CHECK(hipMemset(p, 0x5a, bytes));
    printf("hipMalloc alone again (%d)                  : %7.1f GB/s\n", rep, time_reads(p, bytes, sink));
    CHECK(hipFree(p));
  }
  CHECK(hipFree(sink));
  return 0;
}
