// Diagnostic micro-benchmark (not part of the product): the MaximizeCell greedy scan of rs_interslice.h alone, one wave per CU,
// sorted records in LDS: the serial form against the vector form (results compared, cycles per scan printed).
//   hipcc --offload-arch=gfx950 -O3 -DMB_R=25 -DMB_S=20 -I../../radiosaber_amd/csrc -I../../include -o mb_greedy mb_greedy.hip && ./mb_greedy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include "rs_interslice.h"

#ifndef MB_R
#define MB_R 25
#endif
#ifndef MB_S
#define MB_S 20
#endif
#define N (MB_R * MB_S)
#ifndef NPROB
#define NPROB 8
#endif

template <int V>
__global__ void __launch_bounds__(64) bench(const uint32_t* recs, const int* quotas, int* out, unsigned long long* cyc, int reps) {
  __shared__ uint32_t s_in[NPROB][N];
  __shared__ uint32_t s_work[N];
  __shared__ RsMisc misc;
  __shared__ int s_quota[NPROB][MB_S];
  const int lane = threadIdx.x & 63;
  for (int p = 0; p < NPROB; ++p) {
    for (int i = lane; i < N; i += 64) s_in[p][i] = recs[p * N + i];
    if (lane < MB_S) s_quota[p][lane] = quotas[p * MB_S + lane];
  }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  int acc = 0;
  for (int r = 0; r < reps; ++r)
    for (int p = 0; p < NPROB; ++p) {
      if (lane < MB_S) misc.quota[lane] = s_quota[p][lane];
      for (int i = lane; i < N; i += 64) s_work[i] = s_in[p][i];
      int got = 0, ms = 0;
      if (V == 0) ms = interslice_maximize_cell<MB_S, MB_R>(s_work, &misc, MB_S, MB_R, got);
      else if (V == 1) ms = interslice_maximize_cell_vector<MB_S, MB_R, (MB_S <= 32 && MB_R <= 32)>(s_work, &misc, MB_S, MB_R, got);
      else if (V == 2) ms = interslice_maximize_cell<0, 0>(s_work, &misc, MB_S, MB_R, got);
      else if (V == 3) ms = interslice_maximize_cell_vector<0, 0, (MB_S <= 32 && MB_R <= 32)>(s_work, &misc, MB_S, MB_R, got);
      else ms = s_work[(lane * 7) % N]; /* the copy alone */
      acc += ms * 3 + got;
      if (r == 0 && blockIdx.x == 0) {
        out[p * 128 + lane] = lane < MB_R ? ms : -2;
        out[p * 128 + 64 + lane] = lane < MB_S ? got : -2;
      }
    }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x7fffffff) out[0] = acc;
}

int main() {
  std::mt19937 g(7);
  std::vector<uint32_t> recs(NPROB * N);
  std::vector<int> quotas(NPROB * MB_S);
  for (int p = 0; p < NPROB; ++p) { /* keys skewed towards the top like MaximizeCell's winners */
    std::vector<uint32_t> v;
    for (int r = 0; r < MB_R; ++r)
      for (int s = 0; s < MB_S; ++s) {
        int k = 15 - (int)(std::abs(std::normal_distribution<double>(0, 2.5)(g)));
        if (k < 1) k = 1;
        v.push_back((uint32_t)k << 16 | r << 8 | s);
      }
    std::shuffle(v.begin(), v.end(), g);
    std::stable_sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });
    std::copy(v.begin(), v.end(), recs.begin() + p * N);
    for (int s = 0; s < MB_S; ++s) quotas[p * MB_S + s] = MB_R / MB_S;
    for (int k = 0; k < MB_R - MB_S * (MB_R / MB_S); ++k) quotas[p * MB_S + g() % MB_S]++;
  }
  uint32_t* d_recs; int* d_q; int* d_out; unsigned long long* d_cyc;
  (void)hipMalloc(&d_recs, recs.size() * 4); (void)hipMalloc(&d_q, quotas.size() * 4);
  (void)hipMalloc(&d_out, NPROB * 128 * 4); (void)hipMalloc(&d_cyc, 256 * 8);
  (void)hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_q, quotas.data(), quotas.size() * 4, hipMemcpyHostToDevice);
  const int reps = 100;
  const char* names[] = {"serial, shape known", "vector, shape known", "serial, run-time shape", "vector, run-time shape", "copy only"};
  std::vector<int> ref;
  for (int v = 0; v < 5; ++v) {
    (void)hipMemset(d_out, 0, NPROB * 128 * 4);
    for (int it = 0; it < 2; ++it) {
      switch (v) {
        case 0: hipLaunchKernelGGL(bench<0>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out, d_cyc, reps); break;
        case 1: hipLaunchKernelGGL(bench<1>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out, d_cyc, reps); break;
        case 2: hipLaunchKernelGGL(bench<2>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out, d_cyc, reps); break;
        case 3: hipLaunchKernelGGL(bench<3>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out, d_cyc, reps); break;
        default: hipLaunchKernelGGL(bench<4>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out, d_cyc, reps); break;
      }
      (void)hipDeviceSynchronize();
    }
    unsigned long long c[256];
    (void)hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    std::vector<int> o(NPROB * 128);
    (void)hipMemcpy(o.data(), d_out, NPROB * 128 * 4, hipMemcpyDeviceToHost);
    int bad = -1;
    if (v == 0) ref = o;
    else if (v < 4) { bad = 0; for (size_t i = 0; i < o.size(); ++i) bad += o[i] != ref[i]; }
    printf("%-24s %7.0f cycles per scan (block 0), %7.0f (block 128)  mismatches vs serial: %d\n", names[v], (double)c[0] / (reps * NPROB),
           (double)c[128] / (reps * NPROB), bad);
  }
  return 0;
}
