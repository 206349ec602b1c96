// Diagnostic micro-benchmark (not part of the product): what one wave alone on a SIMD sustains on gfx950, in cycles per
// instruction, for the instruction mixes the sort / greedy code is made of -- pure VALU, pure SALU, alternating, mask round
// trips (v_cmp -> s_and -> v_cndmask), ballot -> s_bcnt1 -> v_mbcnt, LDS round trips.  One wave per workgroup, one workgroup per CU
// (and with 2 / 4 waves per SIMD for comparison).
//   hipcc --offload-arch=gfx950 -O3 -o mb_issue mb_issue.hip && ./mb_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int V>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, int* sink) {
  __shared__ int lds[2048];
  const int lane = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 1024] = threadIdx.x;
  __syncthreads();
  int a = lane, b = lane * 3, c = lane ^ 5, d = lane + 7;
  int sa = iters, sb = iters * 3;
  unsigned long long sx = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) { /* 64 independent VALU (4 chains of 16) */
      REP16(asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
    } else if (V == 1) { /* 64 dependent VALU */
      REP64(asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));)
    } else if (V == 2) { /* 64 dependent SALU */
      REP64(asm volatile("s_add_i32 %0, %0, 1" : "+s"(sa) : : "scc");)
    } else if (V == 3) { /* 64 SALU, two chains */
      REP16(asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %1, %1, 1" : "+s"(sa), "+s"(sb) : : "scc");)
    } else if (V == 4) { /* alternating independent VALU / SALU: 32 + 32 */
      REP16(asm volatile("v_add_u32 %0, %0, 1\n s_add_i32 %2, %2, 1\n v_add_u32 %1, %1, 1\n s_add_i32 %3, %3, 1" : "+v"(a), "+v"(b), "+s"(sa), "+s"(sb) : : "scc");)
    } else if (V == 5) { /* mask round trip: v_cmp -> s_and -> v_cndmask (dependent), 16 x 4 instructions */
      REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_u32 %1, %1, 1" : "+v"(a), "+v"(b) : : "vcc", "scc");)
    } else if (V == 6) { /* ballot -> s_bcnt1 -> v_mbcnt lo/hi -> v_add (dependent), 16 x 5 instructions, counted as 64 + 16 */
      REP16(asm volatile("v_cmp_lt_i32 vcc, %0, %1\n s_bcnt1_i32_b64 %2, vcc\n v_mov_b32 %3, %2\n v_mbcnt_lo_u32_b32 %0, vcc_lo, %3\n v_mbcnt_hi_u32_b32 %0, vcc_hi, %0\n v_add_u32 %1, %1, %0" : "+v"(a), "+v"(b), "+s"(sa), "+v"(c) : : "vcc", "scc");)
    } else if (V == 7) { /* dependent LDS round trips: 16 x (ds_read + wait + v_and) */
      REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0\n v_add_u32 %1, %1, %0" : "+v"(a), "+v"(b));)
    } else if (V == 8) { /* ds_bpermute round trips, dependent: 16 x */
      REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xfc, %0\n v_add_u32 %1, %1, 4" : "+v"(a), "+v"(b));)
    } else if (V == 9) { /* taken branches: 16 x (s_cmp + s_cbranch taken) + VALU */
      REP16(asm volatile("s_cmp_lg_u32 %1, 0x7fffffff\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, 7\n 1:\n v_add_u32 %0, %0, 1\n s_add_i32 %1, %1, 0" : "+v"(a), "+s"(sa) : : "scc");)
    } else if (V == 10) { /* exec save / restore around one VALU: 16 x 4 */
      REP16(asm volatile("v_cmp_le_i32 vcc, %0, %0\n s_and_saveexec_b64 %2, vcc\n v_add_u32 %0, %0, 1\n s_or_b64 exec, exec, %2" : "+v"(a), "+v"(b), "=&s"(sx) : : "vcc", "scc");)
    } else if (V == 11) { /* v_readlane (VALU -> SGPR) then SALU use then VALU use: 16 x 3 dependent */
      REP16(asm volatile("v_readlane_b32 %1, %0, 3\n s_add_i32 %1, %1, 1\n v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, 1" : "+v"(a), "+s"(sa), "+v"(b) : : "scc");)
    } else if (V == 12) { /* DPP dependent chain: 64 x v_add row_shr:1 */
      REP64(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));)
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  if (a + b + c + d + sa + sb + (int)sx == 0x12345678) sink[0] = a;
}

template <int V>
static void run(const char* name, int n_instr, unsigned long long* d_out, int* d_sink) {
  const int iters = 200;
  printf("%-64s", name); fflush(stdout);
  for (int threads : {64, 256, 512, 1024}) { /* 1, 1, 2, 4 waves per SIMD */
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, d_out, iters, d_sink);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(threads), 0, 0, d_out, iters, d_sink);
    (void)hipDeviceSynchronize();
    unsigned long long c[16];
    (void)hipMemcpy(c, d_out, sizeof c, hipMemcpyDeviceToHost);
    printf("  %5.2f", (double)c[0] / iters / n_instr); fflush(stdout);
  }
  printf("\n");
}

int main() {
  unsigned long long* d_out; int* d_sink;
  (void)hipMalloc(&d_out, 256 * 16 * 8); (void)hipMalloc(&d_sink, 64);
  printf("cycles per instruction seen by one wave; workgroup = 1 | 4 | 8 | 16 waves (1 | 1 | 2 | 4 per SIMD), one workgroup per CU\n");
  run<0>("independent v_add (4 chains)", 64, d_out, d_sink);
  run<1>("dependent v_add", 64, d_out, d_sink);
  run<2>("dependent s_add", 64, d_out, d_sink);
  run<3>("s_add, two chains", 64, d_out, d_sink);
  run<4>("alternating independent v_add / s_add", 64, d_out, d_sink);
  run<5>("v_cmp -> s_and vcc -> v_cndmask -> v_add (per instruction)", 64, d_out, d_sink);
  run<6>("v_cmp -> s_bcnt1 -> v_mov -> v_mbcnt_lo -> v_mbcnt_hi -> v_add (per instr.)", 96, d_out, d_sink);
  run<7>("dependent ds_read_b32 round trip (per trip, 4 instr.)", 16, d_out, d_sink);
  run<8>("dependent ds_bpermute round trip (per trip, 4 instr.)", 16, d_out, d_sink);
  run<9>("s_cmp + taken s_cbranch + v_add + s_add (per group of 4 executed)", 16, d_out, d_sink);
  run<10>("v_cmp, s_and_saveexec, v_add, s_or exec (per group of 4)", 16, d_out, d_sink);
  run<11>("v_readlane -> s_add -> v_add (+1 indep. v_add) (per group of 4)", 16, d_out, d_sink);
  run<12>("dependent v_add_dpp row_shr:1", 64, d_out, d_sink);
  return 0;
}
