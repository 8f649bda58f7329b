// H2D of a page-cached, memory-mapped file: one thread vs several threads copying slices on their own streams,
// plain (pageable) copies vs register + async copy.  usage: h2d_file <path>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv)
{
  CK(hipSetDevice(0));
  CK(hipFree(0));
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  fstat(fd, &st);
  const size_t N = st.st_size;
  void *dev;
  CK(hipMalloc(&dev, N + 64));
  for (int mode = 0; mode < 2; ++mode)
    for (int th : {1, 2, 4, 8})
    {
      const char *src = (const char *) mmap(nullptr, N, PROT_READ, MAP_PRIVATE, fd, 0);
      const size_t piece = 32u << 20;
      const size_t np = (N + piece - 1) / piece;
      double t0 = now();
      std::vector<std::thread> ts;
      for (int i = 0; i < th; ++i)
        ts.emplace_back([&, i] {
          CK(hipSetDevice(0));
          hipStream_t s;
          CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
          for (size_t p = i; p < np; p += th)
          {
            const size_t lo = p * piece, n = std::min(piece, N - lo);
            if (mode == 0)
              CK(hipMemcpyAsync((char *) dev + lo, src + lo, n, hipMemcpyHostToDevice, s));
            else
            {
              CK(hipHostRegister((void *) (src + lo), n, hipHostRegisterDefault));
              CK(hipMemcpyAsync((char *) dev + lo, src + lo, n, hipMemcpyHostToDevice, s));
            }
          }
          CK(hipStreamSynchronize(s));
          if (mode == 1)
            for (size_t p = i; p < np; p += th) CK(hipHostUnregister((void *) (src + p * piece)));
          CK(hipStreamDestroy(s));
        });
      for (auto &t : ts) t.join();
      double t1 = now();
      printf("%s, %d threads: %.2f ms = %.1f GB/s\n", mode ? "register + async copy" : "pageable copy", th, (t1 - t0) * 1e3, N / (t1 - t0) / 1e9);
      munmap((void *) src, N);
    }
  return 0;
}
