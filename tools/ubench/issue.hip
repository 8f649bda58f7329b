// micro-benchmark: what does a wavefront that is ALONE on its SIMD pay per instruction?  Each case is a block of 64 copies of
// a short pattern inside a loop, timed with s_memtime; printed as shader clocks per instruction.
// hipcc --offload-arch=gfx950 -O2 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
#define CASE(ID, NINSTR, BODY)                                                                                      \
  if (which == ID)                                                                                                  \
  {                                                                                                                 \
    unsigned long long t0 = __builtin_readcyclecounter();                                                           \
    for (int it = 0; it < iters; ++it) asm volatile(R64(BODY) ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "s20", "s21", "s22", "s23", "s24", "s25", "vcc", "scc", "memory"); \
    unsigned long long t1 = __builtin_readcyclecounter();                                                           \
    if (threadIdx.x == 0) out[0] = (t1 - t0) * 100 / ((unsigned long long) iters * 64 * NINSTR);                    \
  }
// the same with EXEC narrowed to the given lanes for the timed loop (does the SIMD skip passes over idle lanes?)
#define CASEX(ID, NINSTR, LO, HI, BODY)                                                                             \
  if (which == ID)                                                                                                  \
  {                                                                                                                 \
    asm volatile("s_mov_b64 s[26:27], exec\n s_mov_b32 exec_lo, " #LO "\n s_mov_b32 exec_hi, " #HI "\n" ::: "s26", "s27", "memory"); \
    unsigned long long t0 = __builtin_readcyclecounter();                                                           \
    for (int it = 0; it < iters; ++it) asm volatile(R64(BODY) ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "s20", "s21", "s22", "s23", "s24", "s25", "vcc", "scc", "memory"); \
    unsigned long long t1 = __builtin_readcyclecounter();                                                           \
    asm volatile("s_mov_b64 exec, s[26:27]\n" ::: "memory");                                                        \
    if (threadIdx.x == 0) out[0] = (t1 - t0) * 100 / ((unsigned long long) iters * 64 * NINSTR);                    \
  }
__global__ __launch_bounds__(64) void k(unsigned long long *out, int which, int iters)
{
  __shared__ uint32_t lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  asm volatile("v_mov_b32 v1, 1\n v_mov_b32 v2, 2\n v_mov_b32 v3, 3\n v_mov_b32 v4, 4\n v_mov_b32 v5, 0\n v_mov_b32 v6, 64\n s_mov_b64 s[20:21], 5\n s_mov_b32 s22, 7\n" ::: "v1", "v2", "v3", "v4", "v5", "v6", "s20", "s21", "s22");
  CASE(0, 1, "v_add_u32 v1, v2, v3\n")                                            // independent VALU
  CASE(1, 1, "v_add_u32 v1, v1, v2\n")                                            // dependent VALU chain
  CASE(2, 2, "v_add_u32 v1, v1, v2\n v_add_u32 v3, v3, v2\n")                     // two interleaved chains
  CASE(3, 1, "v_cndmask_b32 v1, v2, v3, vcc\n")                                   // select on vcc, independent
  CASE(4, 2, "v_cmp_ge_u32 vcc, v1, v2\n v_cndmask_b32 v1, v2, v3, vcc\n")        // compare -> select (vcc dependency)
  CASE(5, 2, "v_cmp_ge_u32_e64 s[20:21], v1, v2\n v_cndmask_b32_e64 v1, v2, v3, s[20:21]\n")  // the same through an SGPR pair
  CASE(6, 1, "v_cmp_ge_u32_sdwa vcc, v1, v2 src0_sel:WORD_1 src1_sel:WORD_1\n")   // SDWA compare
  CASE(7, 1, "v_lshl_add_u32 v1, v2, 3, s22\n")                                   // VOP3 with an SGPR operand
  CASE(8, 1, "s_add_u32 s22, s22, 1\n")                                           // dependent SALU
  CASE(9, 2, "v_add_u32 v1, v1, v2\n s_add_u32 s22, s22, 1\n")                    // VALU / SALU alternating
  CASE(10, 2, "v_cmp_eq_u32_e64 s[20:21], v1, v2\n s_and_b64 s[24:25], s[20:21], exec\n")  // VALU writes SGPR, SALU reads it
  CASE(11, 1, "s_waitcnt lgkmcnt(0)\n")                                           // a wait with nothing outstanding
  CASE(12, 1, "s_cbranch_scc1 0\n")                                               // branch, never taken (offset 0 = next instruction)
  CASE(13, 1, "ds_write_b32 v5, v1\n")                                            // LDS store issue
  CASE(14, 2, "ds_read2_b32 v[8:9], v5 offset1:1\n s_waitcnt lgkmcnt(0)\n")       // LDS read round trip (read + wait)
  CASE(15, 2, "ds_read_b32 v8, v5\n s_waitcnt lgkmcnt(0)\n")
  CASE(16, 2, "ds_read_b64 v[8:9], v5\n s_waitcnt lgkmcnt(0)\n")
  CASE(17, 2, "ds_read_b128 v[6:9], v5\n s_waitcnt lgkmcnt(0)\n")
  CASE(18, 3, "s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[20:21]\n s_mov_b64 exec, s[24:25]\n")  // exec switching
  CASE(19, 1, "v_addc_co_u32_e32 v1, vcc, v1, v1, vcc\n")
  CASE(20, 4, "ds_read2_b32 v[8:9], v5 offset1:1\n v_add_u32 v1, v2, v3\n v_add_u32 v4, v2, v3\n s_waitcnt lgkmcnt(0)\n")
  CASE(21, 2, "s_setprio 3\n s_setprio 0\n")
  CASEX(22, 1, 0xffffffff, 0, "v_add_u32 v1, v1, v2\n")
  CASEX(23, 1, 0xffff, 0, "v_add_u32 v1, v1, v2\n")
  CASEX(24, 1, 0xff, 0, "v_add_u32 v1, v1, v2\n")
  CASEX(25, 1, 0xff, 0, "v_cmp_ge_u32_sdwa vcc, v1, v2 src0_sel:WORD_1 src1_sel:WORD_1\n")
  CASEX(26, 2, 0xff, 0, "ds_read2_b32 v[8:9], v5 offset1:1\n s_waitcnt lgkmcnt(0)\n")
  CASEX(27, 1, 0xff, 0, "ds_write_b32 v5, v1\n")
  CASEX(28, 1, 0xff, 0, "v_lshl_add_u32 v1, v2, 3, s22\n")
  CASEX(29, 2, 0xffff, 0, "ds_read2_b32 v[8:9], v5 offset1:1\n s_waitcnt lgkmcnt(0)\n")
  if (threadIdx.x == 99) out[1] = lds[threadIdx.x];
}
int main()
{
  unsigned long long *d, h = 0;
  CK(hipMalloc(&d, 64));
  const char *names[] = {"v_add independent", "v_add dependent chain", "two interleaved v_add chains", "v_cndmask (vcc) independent", "v_cmp -> v_cndmask via vcc", "v_cmp_e64 -> v_cndmask_e64 via SGPR pair",
                         "v_cmp_sdwa", "v_lshl_add with SGPR", "s_add dependent", "v_add / s_add alternating", "v_cmp_e64 -> s_and_b64", "s_waitcnt (nothing outstanding)", "s_cbranch not taken",
                         "ds_write_b32 issue", "ds_read2_b32 + wait", "ds_read_b32 + wait", "ds_read_b64 + wait", "ds_read_b128 + wait", "exec save / set / restore", "v_addc dependent",
                         "ds_read2 + 2 v_add + wait", "s_setprio pair",
                         "v_add dependent, 32 lanes in EXEC", "v_add dependent, 16 lanes in EXEC", "v_add dependent, 8 lanes in EXEC", "v_cmp_sdwa, 8 lanes in EXEC", "ds_read2_b32 + wait, 8 lanes in EXEC",
                         "ds_write_b32 issue, 8 lanes in EXEC", "v_lshl_add with SGPR, 8 lanes in EXEC", "ds_read2_b32 + wait, 16 lanes in EXEC"};
  for (int which = 0; which < 30; ++which)
  {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, which, 200);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, which, 2000);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%-44s %6.2f clocks per instruction\n", names[which], h / 100.0);
  }
  return 0;
}
