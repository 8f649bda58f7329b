// What a chain of small dependent kernels on one stream costs while a persistent kernel runs on another stream, by what the
// persistent workgroups do meanwhile: nothing but polling, plain stores over a private region (dirty lines in the XCDs' L2s),
// write-through (sc1) stores over the same region, loads only.  And by how many other streams hold a small waiting kernel.
// build: hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench/beside.hip -o tools/ubench/beside
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <atomic>
#include <vector>
__global__ void k_small(unsigned *p, unsigned n)
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1u;
}
// mode 0: poll only; 1: plain stores; 2: sc1 stores; 3: plain loads
__global__ __launch_bounds__(256) void k_persist(unsigned *region, unsigned words_per_wg, const unsigned *quit, int mode, unsigned *sink)
{
  unsigned *mine = region + (size_t) blockIdx.x * words_per_wg;
  unsigned acc = 0;
  for (unsigned it = 0;; ++it)
  {
    if (__hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (mode == 0)
      __builtin_amdgcn_s_sleep(64);
    else
      for (unsigned i = threadIdx.x; i < words_per_wg; i += blockDim.x)
      {
        if (mode == 1) mine[i] = it + i;
        else if (mode == 2) __hip_atomic_store(mine + i, it + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else acc += mine[i];
      }
  }
  if (acc == 0x12345u) sink[0] = acc;
}
__global__ void k_wait(const unsigned *quit)
{
  while (__hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(64);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
  const int n_wg = 384;
  const unsigned words = 64 * 1024;  // 256 KB per workgroup
  unsigned *region, *quit, *small, *sink;
  CK(hipMalloc(&region, (size_t) n_wg * words * 4));
  CK(hipMalloc(&quit, 64 * 4));
  CK(hipMalloc(&small, 1 << 22));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(region, 0, (size_t) n_wg * words * 4));
  hipStream_t sp, sc, sq;
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sq, hipStreamNonBlocking));
  std::vector<hipStream_t> sw(12);
  for (auto &s : sw) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto chain = [&](const char *what) {
    for (unsigned n : {256u, 1u << 20})
    {
      const int N = 300;
      CK(hipStreamSynchronize(sc));
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, sc, small, n);
      CK(hipStreamSynchronize(sc));
      auto t1 = std::chrono::steady_clock::now();
      printf("  %-44s chain of %d kernels over %7u words: %6.2f us per kernel\n", what, N, n, std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    }
  };
  chain("idle device");
  for (int waiters : {0, 12})
    for (int mode = 0; mode < 4; ++mode)
    {
      CK(hipMemset(quit, 0, 256));
      hipLaunchKernelGGL(k_persist, dim3(n_wg), dim3(256), 0, sp, region, words, quit, mode, sink);
      for (int w = 0; w < waiters; ++w) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, sw[w], quit);
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
      char what[128];
      snprintf(what, sizeof what, "persistent mode %d (%s), %d waiting kernels", mode, mode == 0 ? "poll" : mode == 1 ? "plain stores" : mode == 2 ? "sc1 stores" : "loads", waiters);
      chain(what);
      const unsigned one = 1;
      CK(hipMemcpyAsync(quit, &one, 4, hipMemcpyHostToDevice, sq));
      CK(hipStreamSynchronize(sq));
      CK(hipStreamSynchronize(sp));
      for (int w = 0; w < waiters; ++w) CK(hipStreamSynchronize(sw[w]));
    }
  // K chains at once, each on its own stream and host thread: what one kernel of a chain costs then
  for (int persist : {0, 1})
  {
    if (persist)
    {
      CK(hipMemset(quit, 0, 256));
      hipLaunchKernelGGL(k_persist, dim3(n_wg), dim3(256), 0, sp, region, words, quit, 0, sink);
    }
    for (int K : {1, 2, 4, 8, 12})
    {
      std::vector<std::thread> th;
      std::vector<double> per(K);
      std::atomic<int> ready{0};
      for (int t = 0; t < K; ++t)
        th.emplace_back([&, t] {
          hipSetDevice(0);
          const int N = 300;
          unsigned *mine = small + (size_t) t * 65536;
          hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, sw[t], mine, 256u);
          hipStreamSynchronize(sw[t]);
          ++ready;
          while (ready.load() < K) {}
          auto t0 = std::chrono::steady_clock::now();
          for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, sw[t], mine, 256u);
          hipStreamSynchronize(sw[t]);
          auto t1 = std::chrono::steady_clock::now();
          per[t] = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        });
      for (auto &x : th) x.join();
      double a = 0, mx = 0;
      for (double v : per) { a += v / K; mx = v > mx ? v : mx; }
      printf("  %2d chains at once%s: %6.2f us per kernel of a chain (slowest chain %6.2f) = one kernel per %5.2f us overall\n", K, persist ? " beside a polling persistent kernel" : "", a, mx, a / K);
    }
    if (persist)
    {
      const unsigned one = 1;
      CK(hipMemcpyAsync(quit, &one, 4, hipMemcpyHostToDevice, sq));
      CK(hipStreamSynchronize(sq));
      CK(hipStreamSynchronize(sp));
    }
  }
  return 0;
}
