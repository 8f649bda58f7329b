// H2D feed micro-benchmark: cost of pinned allocations, pageable vs pinned copies, threaded memcpy into pinned memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  hipSetDevice(0);
  hipFree(0);
  const size_t N = 256u << 20;
  char *src = (char *) malloc(N);
  memset(src, 1, N);
  void *dev;
  hipMalloc(&dev, N);
  for (size_t mb : {16, 64, 256})
  {
    double t0 = now();
    void *p;
    hipHostMalloc(&p, mb << 20, hipHostMallocDefault);
    double t1 = now();
    memset(p, 0, mb << 20);
    double t2 = now();
    hipHostFree(p);
    printf("hipHostMalloc %zu MB: %.2f ms (first touch %.2f ms)\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  }
  for (int rep = 0; rep < 2; ++rep)
  {
    double t0 = now();
    hipMemcpy(dev, src, N, hipMemcpyHostToDevice);
    double t1 = now();
    printf("pageable H2D 256 MB: %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  }
  void *pin;
  hipHostMalloc(&pin, N, hipHostMallocDefault);
  for (int th : {1, 2, 4, 8, 16})
  {
    double t0 = now();
    std::vector<std::thread> ts;
    for (int i = 0; i < th; ++i) ts.emplace_back([&, i] { memcpy((char *) pin + N / th * i, src + N / th * i, N / th); });
    for (auto &t : ts) t.join();
    double t1 = now();
    printf("memcpy into pinned, %d threads: %.2f ms = %.1f GB/s\n", th, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  }
  for (int rep = 0; rep < 2; ++rep)
  {
    double t0 = now();
    hipMemcpy(dev, pin, N, hipMemcpyHostToDevice);
    double t1 = now();
    printf("pinned H2D 256 MB: %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  }
  double t0 = now();
  hipHostRegister(src, N, hipHostRegisterDefault);
  double t1 = now();
  printf("hipHostRegister 256 MB: %.2f ms\n", (t1 - t0) * 1e3);
  t0 = now();
  hipMemcpy(dev, src, N, hipMemcpyHostToDevice);
  t1 = now();
  printf("registered H2D 256 MB: %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
  return 0;
}
