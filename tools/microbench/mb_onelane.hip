// mb_onelane.hip -- attempt at a stand-alone form of round 4's wrong kernel (profiles/r05_onelane.md): a register-capped body
// (128 VGPRs: __launch_bounds__(512, 4)) with NLIVE per-lane doubles and a few wave-uniform constants live across a wave-uniform
// `if` behind a barrier (the shape of the quota phase: one wave works, the others skip it with EXEC = 0), then uses of everything.
// The defect needs the allocator to pick the join block's head as a spill / split point:
//   for n in 24 32 40 48 56 64; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DNLIVE=$n [-DHEAVY] --cuda-device-only -S -o /tmp/mb_$n.s \
//       tools/microbench/mb_onelane.hip && python tools/lint_exec_restore.py /tmp/mb_$n.s; done
// finds nothing with this toolchain (the listings spill -- up to 92 stores, 420 B of scratch -- but every one sits inside its block);
// the reproducer that does show it is tools/onelane_repro.sh.
#include <hip/hip_runtime.h>
#ifndef NLIVE
#define NLIVE 40
#endif

__global__ void __launch_bounds__(512, 4) mb_onelane(const double* __restrict__ in, double* __restrict__ out, int n_iter, int S) {
  __shared__ double sh[1024];
  __shared__ int quota[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  double a[NLIVE];
#pragma unroll
  for (int i = 0; i < NLIVE; ++i) a[i] = in[(size_t)blockIdx.x * blockDim.x * NLIVE + (size_t)i * blockDim.x + tid];
  double least = 1.7976931348623157e308, acc = 0;
  for (int it = 0; it < n_iter; ++it) {
    sh[tid] = a[0] + acc;
    sh[tid + 512] = a[7];
    __syncthreads();
    if (wave == nwaves - 1) { /* one wave works (lanes = slices), the others arrive at the join with EXEC = 0 */
      int q = lane < S ? (int)(sh[lane] * 3.0) : 0;
#ifdef HEAVY
      double w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = sh[(lane * 9 + k * 67) & 1023] / (1.0 + sh[(lane + k) & 1023]);
#pragma unroll
      for (int k = 0; k < 8; ++k) q += (int)(w[k] * (double)(k + 1));
#endif
      for (int d = 1; d < 64; d <<= 1) q += __shfl_up(q, d) * (lane >= d ? 1 : 0);
      if (lane < S) quota[lane] = q;
    }
    __syncthreads();
    if (wave == 0) { /* the wave that skipped the region uses the constants */
      double best = least;
      for (int k = 0; k < S; ++k) {
        const double loss = sh[(lane + k) & 1023] - sh[512 + ((lane * 3 + k) & 511)];
        if (loss < best && quota[k] > 0) best = loss;
      }
      acc += best == least ? 0.0 : best;
    }
#pragma unroll
    for (int i = 0; i < NLIVE; ++i) a[i] = a[i] * 0.98 + (double)quota[(lane + i) & 63] * 0.02;
    __syncthreads();
  }
  double s = acc;
#pragma unroll
  for (int i = 0; i < NLIVE; ++i) s += a[i];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}
