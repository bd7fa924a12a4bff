// runtime_call_cost.hip — host cost of the HIP runtime calls a sharded search makes per shard (one GPU, g streams):
// hipSetDevice, hipStreamWaitEvent, a small kernel launch, hipEventRecord; issued from ONE thread, and the same
// per-shard sequence issued from g threads at once (does the runtime serialise them?).
//   hipcc --offload-arch=gfx950 -O2 -o benchmarks/runtime_call_cost benchmarks/runtime_call_cost.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main() {
  const int g = 8, iters = 2000;
  CK(hipSetDevice(0));
  hipStream_t main_st; CK(hipStreamCreateWithFlags(&main_st, hipStreamNonBlocking));
  std::vector<hipStream_t> st(g); std::vector<hipEvent_t> done(g);
  hipEvent_t evq; CK(hipEventCreateWithFlags(&evq, hipEventDisableTiming));
  int* buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
  for (int s = 0; s < g; ++s) { CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&done[s], hipEventDisableTiming)); }
  double t_set = 0, t_wait = 0, t_launch = 0, t_rec = 0, t_mwait = 0, t_total = 0;
  for (int it = -200; it < iters; ++it) {
    auto a0 = clk::now();
    CK(hipEventRecord(evq, main_st));
    auto a1 = clk::now();
    double ts = 0, tw = 0, tl = 0, tr = 0;
    for (int s = 0; s < g; ++s) {
      auto b0 = clk::now(); CK(hipSetDevice(0));
      auto b1 = clk::now(); CK(hipStreamWaitEvent(st[s], evq, 0));
      auto b2 = clk::now(); hipLaunchKernelGGL(tiny, dim3(512), dim3(256), 0, st[s], buf + 64 * s);
      auto b3 = clk::now(); CK(hipEventRecord(done[s], st[s]));
      auto b4 = clk::now();
      ts += us(b0, b1); tw += us(b1, b2); tl += us(b2, b3); tr += us(b3, b4);
    }
    auto a2 = clk::now();
    for (int s = 0; s < g; ++s) CK(hipStreamWaitEvent(main_st, done[s], 0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, main_st, buf + 1000);
    auto a3 = clk::now();
    if (it >= 0) { t_set += ts; t_wait += tw; t_launch += tl; t_rec += tr; t_mwait += us(a2, a3); t_total += us(a0, a3); }
    if ((it & 63) == 63) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  printf("{\"one_thread\": {\"per_shard_us\": {\"hipSetDevice\": %.2f, \"hipStreamWaitEvent\": %.2f, \"launch\": %.2f, \"hipEventRecord\": %.2f}, "
         "\"main_8_waits_plus_merge_launch_us\": %.2f, \"total_us_per_8_shard_call\": %.2f},\n",
         t_set / iters / g, t_wait / iters / g, t_launch / iters / g, t_rec / iters / g, t_mwait / iters, t_total / iters);
  // the same per-shard sequence from g threads at once (spinning workers, one stream each)
  std::atomic<int> seq{0}, fin{0}; std::atomic<bool> stop{false};
  std::vector<std::thread> th;
  for (int s = 0; s < g; ++s) th.emplace_back([&, s]() {
    (void)hipSetDevice(0);
    int seen = 0;
    while (true) {
      while (seq.load(std::memory_order_acquire) == seen) { if (stop.load()) return; }
      ++seen;
      (void)hipStreamWaitEvent(st[s], evq, 0);
      hipLaunchKernelGGL(tiny, dim3(512), dim3(256), 0, st[s], buf + 64 * s);
      (void)hipEventRecord(done[s], st[s]);
      fin.fetch_add(1, std::memory_order_release);
    }
  });
  double t_thr = 0, t_issue = 0;
  for (int it = -200; it < iters; ++it) {
    auto a0 = clk::now();
    CK(hipEventRecord(evq, main_st));
    fin.store(0); seq.fetch_add(1, std::memory_order_release);
    while (fin.load(std::memory_order_acquire) < g) {}
    auto a1 = clk::now();
    for (int s = 0; s < g; ++s) CK(hipStreamWaitEvent(main_st, done[s], 0));
    hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, main_st, buf + 1000);
    auto a2 = clk::now();
    if (it >= 0) { t_issue += us(a0, a1); t_thr += us(a0, a2); }
    if ((it & 63) == 63) CK(hipDeviceSynchronize());
  }
  stop.store(true);
  for (auto& t : th) t.join();
  CK(hipDeviceSynchronize());
  printf(" \"eight_threads\": {\"record_wake_issue_us\": %.2f, \"total_us_per_8_shard_call\": %.2f}}\n", t_issue / iters, t_thr / iters);
  return 0;
}
