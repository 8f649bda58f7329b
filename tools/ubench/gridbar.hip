// Latency of a grid-wide barrier inside one persistent kernel (monotonic counter in global memory, agent-scope release / acquire)
// for G co-resident workgroups of 256 threads, with a dependent store -> load across workgroups in every phase (so the fences
// have something to do).  Compare with a dependent kernel launch (tools/ubench/graph.hip: 3.4 us, 2.0 us from a graph).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gridbar.hip -o tools/ubench/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void grid_barrier(uint32_t *bar, uint32_t G, uint32_t &epoch)
{
  __syncthreads();
  if (threadIdx.x == 0)
  {
    ++epoch;
    __atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE);  // agent scope is the default for __atomic builtins on global memory
    const uint32_t target = epoch * G;
    while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
}

__global__ __launch_bounds__(256) void k_bar(uint32_t *bar, uint32_t *data, uint32_t G, uint32_t rounds, uint32_t *bad)
{
  uint32_t epoch = 0;
  const uint32_t b = blockIdx.x;
  for (uint32_t r = 0; r < rounds; ++r)
  {
    // every workgroup writes a value its right neighbour checks after the barrier
    data[b * 256 + threadIdx.x] = r * 7919u + b + threadIdx.x;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    grid_barrier(bar, G, epoch);
    const uint32_t nb = (b + 1) % G;
    if (data[nb * 256 + threadIdx.x] != r * 7919u + nb + threadIdx.x) atomicAdd(bad, 1u);
    grid_barrier(bar, G, epoch);
  }
}

int main()
{
  uint32_t *bar, *data, *bad;
  hipMalloc(&bar, 4);
  hipMalloc(&bad, 4);
  hipMalloc(&data, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (uint32_t G : {1u, 8u, 32u, 64u, 128u, 256u, 512u})
  {
    const uint32_t rounds = 2000;
    hipMemset(bar, 0, 4);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, 0, bar, data, G, 10u, bad);
    hipDeviceSynchronize();
    hipMemset(bar, 0, 4);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, 0, bar, data, G, rounds, bad);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t hb = 0;
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("G=%3u workgroups: %.2f us per barrier (+ a dependent store/load), stale reads %u\n", G, ms * 1e3 / (2.0 * rounds), hb);
  }
  return 0;
}
