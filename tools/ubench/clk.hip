// Does a lone wavefront run faster or slower when the rest of the chip is busy?  (clock / power management probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_lds_rw_chain(uint32_t *out, int iters)
{
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (i * 97 + 13) & 4095;
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (int i = 0; i < iters; ++i)
  {
    uint32_t q = s[p];
    s[p] = (q + 64) & 4095;
    p = (q & 4032) | threadIdx.x;
  }
  out[threadIdx.x] = p;
}
__global__ __launch_bounds__(256) void k_busy_alu(float *out, int iters)
{
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i)
  {
#pragma unroll
    for (int j = 0; j < 32; ++j) a = a * b + 0.5f;
  }
  if (a == 12345.f) out[0] = a;
}
__global__ __launch_bounds__(256) void k_busy_mem(const float4 *in, float4 *out, size_t n, int reps)
{
  float4 acc = make_float4(0, 0, 0, 0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
    {
      float4 v = in[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  if (acc.x == 12345.f) out[0] = acc;
}

int main()
{
  uint32_t *out; float *fo; float4 *big, *bo;
  const size_t N = (size_t) 1 << 28;  // 4 GiB of float4
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&fo, 4096)); CK(hipMalloc(&big, N * 16)); CK(hipMalloc(&bo, 64));
  CK(hipMemset(big, 0, N * 16));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 400000;
  float ms;
  for (int mode = 0; mode < 4; ++mode)
  {
    for (int rep = 0; rep < 2; ++rep)
    {
      if (mode == 1) hipLaunchKernelGGL(k_busy_alu, dim3(255 * 4), dim3(256), 0, s2, fo, 300000);      // ~all other CUs, ALU bound
      if (mode == 2) hipLaunchKernelGGL(k_busy_mem, dim3(255 * 4), dim3(256), 0, s2, big, bo, N, 12);   // HBM bound
      if (mode == 3) hipLaunchKernelGGL(k_busy_alu, dim3(32), dim3(256), 0, s2, fo, 300000);           // a little company
      CK(hipEventRecord(a, s1));
      hipLaunchKernelGGL(k_lds_rw_chain, dim3(1), dim3(64), 0, s1, out, iters);
      CK(hipEventRecord(b, s1));
      CK(hipEventSynchronize(b));
      CK(hipEventElapsedTime(&ms, a, b));
      CK(hipDeviceSynchronize());
      const char *nm[4] = {"alone", "with ALU load on the other CUs", "with HBM streaming on the other CUs", "with 32 ALU blocks"};
      printf("lone-wave LDS load+store chain %-40s: %.1f ns per step\n", nm[mode], ms * 1e6 / iters);
    }
  }
  return 0;
}
