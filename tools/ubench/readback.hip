// What a lane of bk_mask_and_cluster pays for "launch a small kernel, read 16 bytes back, decide": per iteration, for 1 / 2 / 4 / 8
// host threads doing the same on their own streams at once, with the destination in pageable memory, in pinned memory, and with
// the kernel storing straight into mapped pinned memory (no copy command at all).
// build: hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench/readback.hip -o tools/ubench/readback
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_small(unsigned *dev, unsigned *mapped, unsigned v)
{
  if (threadIdx.x == 0)
  {
    dev[0] = v;
    if (mapped) { mapped[0] = v; }
  }
}

static double run(int threads, int mode, int iters)
{
  std::vector<std::thread> th;
  std::vector<double> us(threads, 0.0);
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&, t] {
      hipSetDevice(0);
      hipStream_t st;
      hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
      unsigned *dev, *pinned, *mapped_dev = nullptr;
      hipMalloc(&dev, 64);
      hipHostMalloc(&pinned, 64, hipHostMallocMapped);
      hipHostGetDevicePointer((void **) &mapped_dev, pinned, 0);
      unsigned pageable[4];
      for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, dev, (unsigned *) nullptr, 1u); hipStreamSynchronize(st); }
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < iters; ++i)
      {
        if (mode == 0) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, dev, (unsigned *) nullptr, (unsigned) i); hipMemcpyAsync(pageable, dev, 16, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        if (mode == 1) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, dev, (unsigned *) nullptr, (unsigned) i); hipMemcpyAsync(pinned, dev, 16, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
        if (mode == 2) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st, dev, mapped_dev, (unsigned) i); hipStreamSynchronize(st); if (((volatile unsigned *) pinned)[0] != (unsigned) i) printf("stale\n"); }
      }
      us[t] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
      hipFree(dev);
      hipHostFree(pinned);
      hipStreamDestroy(st);
    });
  for (auto &x : th) x.join();
  double m = 0;
  for (double v : us) m = v > m ? v : m;
  return m;
}

int main(int argc, char **argv)
{
  if (argc > 1) hipSetDeviceFlags(atoi(argv[1]) == 1 ? hipDeviceScheduleSpin : atoi(argv[1]) == 2 ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync);
  printf("flags arg %s, hardware threads %u\n", argc > 1 ? argv[1] : "-", std::thread::hardware_concurrency());
  const char *names[3] = {"kernel + D2H 16 B to pageable memory + sync", "kernel + D2H 16 B to pinned memory + sync  ", "kernel stores to mapped pinned memory + sync"};
  for (int mode = 0; mode < 3; ++mode)
    for (int threads : {1, 4, 5, 6, 8, 12})
      printf("%s, %d thread(s): %.1f us per round trip\n", names[mode], threads, run(threads, mode, 2000));
  return 0;
}
