// Grid barrier without a contended counter: every workgroup stores the round number into its own 4-byte slot, workgroup 0 polls
// all slots (one lane per slot) and publishes the round in a `go` word the others poll.  Agent-scope relaxed accesses for the
// flags, one agent-scope release fence in front of the arrival and one acquire fence behind the departure.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/gridbar2.hip -o tools/ubench/gridbar2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int sleep> __device__ __forceinline__ void grid_barrier2(uint32_t *slots, uint32_t *go, uint32_t G, uint32_t &round)
{
  ++round;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (blockIdx.x == 0)
  {
    for (uint32_t i = threadIdx.x; i < G; i += blockDim.x)
      if (i)
        while (__hip_atomic_load(slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round) __builtin_amdgcn_s_sleep(sleep);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(go, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  else
  {
    if (threadIdx.x == 0)
    {
      __hip_atomic_store(slots + blockIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != round) __builtin_amdgcn_s_sleep(sleep);
    }
    __syncthreads();
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

template <int sleep> __global__ __launch_bounds__(256) void k_bar(uint32_t *slots, uint32_t *go, uint32_t *data, uint32_t G, uint32_t rounds, uint32_t *bad)
{
  uint32_t round = 0;
  const uint32_t b = blockIdx.x;
  for (uint32_t r = 0; r < rounds; ++r)
  {
    data[b * 256 + threadIdx.x] = r * 7919u + b + threadIdx.x;
    grid_barrier2<sleep>(slots, go, G, round);
    const uint32_t nb = (b + 1) % G;
    if (data[nb * 256 + threadIdx.x] != r * 7919u + nb + threadIdx.x) atomicAdd(bad, 1u);
    grid_barrier2<sleep>(slots, go, G, round);
  }
}

int main()
{
  uint32_t *slots, *go, *data, *bad;
  hipMalloc(&slots, 4096 * 4);
  hipMalloc(&go, 256);
  hipMalloc(&bad, 4);
  hipMalloc(&data, 1024 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int sleep : {0, 1, 4})
    for (uint32_t G : {1u, 8u, 32u, 64u, 128u, 256u, 512u})
    {
      const uint32_t rounds = 2000;
      hipMemset(slots, 0, 4096 * 4);
      hipMemset(go, 0, 256);
      hipMemset(bad, 0, 4);
      hipEventRecord(e0, 0);
      if (sleep == 0) hipLaunchKernelGGL(k_bar<0>, dim3(G), dim3(256), 0, 0, slots, go, data, G, rounds, bad);
      if (sleep == 1) hipLaunchKernelGGL(k_bar<1>, dim3(G), dim3(256), 0, 0, slots, go, data, G, rounds, bad);
      if (sleep == 4) hipLaunchKernelGGL(k_bar<4>, dim3(G), dim3(256), 0, 0, slots, go, data, G, rounds, bad);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      uint32_t hb = 0;
      hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      printf("sleep %d G=%3u workgroups: %.2f us per barrier (+ a dependent store/load), stale reads %u\n", sleep, G, ms * 1e3 / (2.0 * rounds), hb);
    }
  return 0;
}
