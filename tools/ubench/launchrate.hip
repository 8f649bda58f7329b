// Host cost of hipLaunchKernelGGL when T threads launch on their own streams at once (no syncs in the loop): a runtime-wide lock
// shows as a per-launch cost that grows with T.
// build: hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench/launchrate.hip -o tools/ubench/launchrate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_nop(unsigned *p) { if (p && threadIdx.x == 999) p[0] = 1; }
int main()
{
  for (int threads : {1, 2, 4, 8})
  {
    std::vector<std::thread> th;
    std::vector<double> issue(threads), total(threads);
    for (int t = 0; t < threads; ++t)
      th.emplace_back([&, t] {
        hipSetDevice(0);
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (unsigned *) nullptr);
        hipStreamSynchronize(st);
        const int N = 20000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st, (unsigned *) nullptr);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        issue[t] = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        total[t] = std::chrono::duration<double, std::micro>(t2 - t0).count() / N;
        hipStreamDestroy(st);
      });
    for (auto &x : th) x.join();
    double a = 0, b = 0;
    for (int t = 0; t < threads; ++t) { a += issue[t] / threads; b += total[t] / threads; }
    printf("%d thread(s): %.2f us of host time per launch, %.2f us per launch until the stream has drained\n", threads, a, b);
  }
  return 0;
}
