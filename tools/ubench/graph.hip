// micro-benchmark: host cost of a chain of small dependent kernels, launched one by one vs replayed as a hipGraph,
// from 1 and 2 host threads (each on its own stream).  hipcc --offload-arch=gfx950 -O2 -pthread graph.hip -o graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void k_tiny(unsigned *p, unsigned n) { unsigned i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct Lane { hipStream_t st; unsigned *buf; hipGraphExec_t ge; };
static void chain(Lane &l, int len, unsigned blocks) { for (int i = 0; i < len; ++i) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(256), 0, l.st, l.buf, blocks * 256); }
int main()
{
  const int LEN = 30, REP = 200;
  for (unsigned blocks : {4u, 512u})
  {
    std::vector<Lane> lanes(2);
    for (Lane &l : lanes)
    {
      CK(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
      CK(hipMalloc(&l.buf, 4 << 20));
      hipGraph_t g;
      CK(hipStreamBeginCapture(l.st, hipStreamCaptureModeThreadLocal));
      chain(l, LEN, blocks);
      CK(hipStreamEndCapture(l.st, &g));
      CK(hipGraphInstantiate(&l.ge, g, nullptr, nullptr, 0));
      CK(hipGraphDestroy(g));
    }
    for (int threads : {1, 2})
      for (int mode : {0, 1})
      {
        auto body = [&](int t) {
          Lane &l = lanes[t];
          for (int r = 0; r < REP; ++r)
          {
            if (mode == 0) chain(l, LEN, blocks); else CK(hipGraphLaunch(l.ge, l.st));
            if (r % 8 == 7) CK(hipStreamSynchronize(l.st));
          }
          CK(hipStreamSynchronize(l.st));
        };
        body(0);  // warm
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(body, t);
        body(0);
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        printf("blocks %u threads %d %s: %.2f us per kernel (wall %.1f ms for %d kernels per thread)\n", blocks, threads, mode ? "graph " : "stream", dt * 1e6 / (REP * LEN), dt * 1e3, REP * LEN);
      }
  }
  // sync round trips: 6 kernels + an 8-byte D2H copy (pageable / pinned destination) + hipStreamSynchronize, from 1, 2, 4 threads
  {
    const int NT = 4, REP2 = 300;
    std::vector<Lane> lanes(NT);
    unsigned long long *pinned;
    CK(hipHostMalloc(&pinned, 64 * NT));
    for (Lane &l : lanes)
    {
      CK(hipStreamCreateWithFlags(&l.st, hipStreamNonBlocking));
      CK(hipMalloc(&l.buf, 4 << 20));
    }
    for (int threads : {1, 2, 4})
      for (int mode : {0, 1, 2})
      {
        auto body = [&](int t) {
          Lane &l = lanes[t];
          unsigned long long stackv = 0;
          for (int r = 0; r < REP2; ++r)
          {
            chain(l, 6, 4);
            if (mode == 0) CK(hipMemcpyAsync(&stackv, l.buf, 8, hipMemcpyDeviceToHost, l.st));
            if (mode == 1) CK(hipMemcpyAsync(pinned + 8 * t, l.buf, 8, hipMemcpyDeviceToHost, l.st));
            CK(hipStreamSynchronize(l.st));
          }
        };
        body(0);
        const double t0 = now();
        std::vector<std::thread> th;
        for (int t = 1; t < threads; ++t) th.emplace_back(body, t);
        body(0);
        for (auto &x : th) x.join();
        const double dt = now() - t0;
        printf("round trips, threads %d, %s: %.1f us per round (6 kernels + copy + sync)\n", threads, mode == 0 ? "pageable D2H" : mode == 1 ? "pinned D2H  " : "sync only   ", dt * 1e6 / REP2);
      }
  }
  return 0;
}
