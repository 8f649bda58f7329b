// single-wave latency probes (gfx950): dependent LDS load chain, LDS load->store->load chain, dependent VALU chain,
// dependent global load chain (L1/L2 resident), scalar branch loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_lds_chain(uint32_t *out, int iters)
{
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i) p = s[p];
  out[threadIdx.x] = p;
}
__global__ __launch_bounds__(64) void k_lds_rw_chain(uint32_t *out, int iters)
{
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i)
  {
    uint32_t q = s[p];
    s[p] = (q + 64) & 4095;   // store depends on load, next load address depends on load
    p = (q & 4032) | threadIdx.x;
  }
  out[threadIdx.x] = p;
}
__global__ __launch_bounds__(64) void k_valu_chain(uint32_t *out, int iters)
{
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int j = 0; j < 16; ++j) p = p * 1664525u + 1013904223u;
  }
  out[threadIdx.x] = p;
}
__global__ __launch_bounds__(64) void k_glb_chain(uint32_t *buf, uint32_t *out, int iters)
{
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i) p = __builtin_nontemporal_load(buf + p) , p = buf[p];
  out[threadIdx.x] = p;
}
__global__ __launch_bounds__(64) void k_glb_chain2(uint32_t *buf, uint32_t *out, int iters)
{
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i) p = buf[p];
  out[threadIdx.x] = p;
}

int main()
{
  uint32_t *out, *buf;
  CK(hipMalloc(&out, 4096));
  const int N = 1 << 14;
  CK(hipMalloc(&buf, N * 4));
  uint32_t *h = new uint32_t[N];
  for (int i = 0; i < N; ++i) h[i] = (i * 97 + 13) & (N - 1);
  CK(hipMemcpy(buf, h, N * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float ms;
  const int iters = 200000;
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(64), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("lds load chain      : %.1f ns per load\n", ms * 1e6 / iters);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k_lds_rw_chain, dim3(1), dim3(64), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("lds load+store chain: %.1f ns per iteration\n", ms * 1e6 / iters);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k_valu_chain, dim3(1), dim3(64), 0, 0, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("valu chain          : %.2f ns per dependent mad (32 per iteration?)\n", ms * 1e6 / iters / 16);
    CK(hipEventRecord(a)); hipLaunchKernelGGL(k_glb_chain2, dim3(1), dim3(64), 0, 0, buf, out, iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("global load chain   : %.1f ns per load (64 KiB working set)\n", ms * 1e6 / iters);
  }
  return 0;
}
