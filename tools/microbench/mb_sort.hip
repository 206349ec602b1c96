// Diagnostic micro-benchmark (not part of the product): the std::sort emulation of rs_sort_device.h alone on MaximizeCell-shaped
// key arrays (tools/sort_study.py dumps them from the CPU oracle): the workgroup-level introsort loop against the task-per-wave
// form, results compared element by element, cycles per sort printed.  One or two workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DRS_TASK_STAMPS -DMB_N=500 -I. -I../../radiosaber_amd/csrc -I../../include -o mb_sort mb_sort.hip
//   ./mb_sort keys_r25.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#ifndef RS_WAVE_FINISH_MAX
#define RS_WAVE_FINISH_MAX 4 /* the task form's hand-over to the packed single-wave finish (the product's kFinishMax at EPT > 1) */
#endif
#include "rs_sort_tasks.h"
#ifdef MB_HYBRID /* round 4: workgroup levels until every sub-range fits MB_HYBRID registers per lane, then tasks (make_hybrid.py) */
#include "rs_sort_hybrid_gen.h"
#endif

#ifndef MB_N
#define MB_N 500
#endif
#ifndef MB_NT
#define MB_NT 512
#endif
#ifndef NPROB
#define NPROB 16
#endif
constexpr int kEpt = (MB_N + MB_NT - 1) / MB_NT;

template <int V>
__global__ void __launch_bounds__(MB_NT) bench(const uint32_t* recs, int n_prob_total, uint32_t* out_loop, uint32_t* out_sorted,
                                                unsigned long long* cyc, int reps) {
  __shared__ uint32_t s_elems[MB_N + 64];
  __shared__ uint32_t s_sorted[MB_N + 64];
  __shared__ int32_t s_cuts[MB_N / 16 + 8];
  __shared__ uint16_t s_ranks[(MB_N + 63) / 64 * 64 + 64];
  __shared__ RsMisc misc;
  const int tid = threadIdx.x;
  unsigned long long t_loop = 0, t_count = 0;
  unsigned long long sub[32];
  for (int i = 0; i < 32; ++i) sub[i] = 0;
  for (int r = 0; r < reps; ++r)
    for (int p0 = 0; p0 < NPROB; ++p0) {
      const int p = (p0 + blockIdx.x * 7) % n_prob_total;
      for (int i = tid; i < MB_N; i += MB_NT) s_elems[i] = recs[(size_t)p * MB_N + i];
      __syncthreads();
      unsigned long long t0 = __builtin_readcyclecounter();
      __builtin_amdgcn_s_setprio(1);
      if (V == 0) introsort_levels_reg<kEpt>(s_elems, MB_N, s_sorted, s_cuts, &misc, nullptr, 0, s_ranks);
#ifdef MB_HYBRID
      else introsort_levels_hybrid<kEpt, MB_HYBRID>(s_elems, MB_N, s_sorted, s_cuts, &misc, sub);
#else
      else introsort_tasks<8, kEpt>(s_elems, MB_N, s_sorted, s_cuts, &misc, 0, sub);
#endif
      unsigned long long t1 = __builtin_readcyclecounter();
      if (r == 0 && blockIdx.x == 0)
        for (int i = tid; i < MB_N; i += MB_NT) out_loop[(size_t)p * MB_N + i] = s_elems[i];
      __builtin_amdgcn_s_setprio(2);
      counting_sort_desc_owned<kEpt>(s_elems, s_sorted, MB_N, &misc);
      __builtin_amdgcn_s_setprio(0);
      unsigned long long t2 = __builtin_readcyclecounter();
      t_loop += t1 - t0;
      t_count += t2 - t1;
      if (r == 0 && blockIdx.x == 0)
        for (int i = tid; i < MB_N; i += MB_NT) out_sorted[(size_t)p * MB_N + i] = s_sorted[i];
      __syncthreads();
    }
  if (tid == 0) { cyc[blockIdx.x * 2] = t_loop; cyc[blockIdx.x * 2 + 1] = t_count; }
  if (tid == 0 && blockIdx.x == 1) for (int i = 0; i < 32; ++i) cyc[4096 + i] = sub[i];
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "keys_r25.bin";
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
  std::vector<uint8_t> keys;
  { uint8_t buf[4096]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) keys.insert(keys.end(), buf, buf + n); }
  fclose(f);
  const int n_prob = (int)(keys.size() / MB_N);
  if (n_prob < NPROB) { fprintf(stderr, "%s holds %d problems of %d keys, need %d\n", path, n_prob, MB_N, NPROB); return 2; }
  std::vector<uint32_t> recs((size_t)n_prob * MB_N);
  for (size_t i = 0; i < recs.size(); ++i) recs[i] = (uint32_t)keys[i] << 16 | (uint32_t)(i % MB_N); /* payload = original position */
  /* CPU reference: the real std::sort with the reference's comparator */
  std::vector<uint32_t> want = recs;
  for (int p = 0; p < n_prob; ++p)
    std::sort(want.begin() + (size_t)p * MB_N, want.begin() + (size_t)(p + 1) * MB_N, [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });
  uint32_t *d_recs, *d_loop, *d_sorted; unsigned long long* d_cyc;
  (void)hipMalloc(&d_recs, recs.size() * 4); (void)hipMalloc(&d_loop, recs.size() * 4); (void)hipMalloc(&d_sorted, recs.size() * 4);
  (void)hipMalloc(&d_cyc, 8192 * 8);
  (void)hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
  const int reps = 20;
#ifdef MB_HYBRID
  const char* names[] = {"workgroup levels", "hybrid -> tasks"};
  printf("N = %d records, %d threads, hybrid hand-over at K <= %d registers per lane\n", MB_N, MB_NT, MB_HYBRID);
#else
  const char* names[] = {"workgroup levels", "task per wave"};
  printf("N = %d records, %d threads\n", MB_N, MB_NT);
#endif
  for (int blocks : {256, 512, 1024})
    for (int v = 0; v < 2; ++v) {
      (void)hipMemset(d_sorted, 0, recs.size() * 4);
      for (int it = 0; it < 2; ++it) {
#define MB_LAUNCH(V_) hipLaunchKernelGGL(bench<V_>, dim3(blocks), dim3(MB_NT), 0, 0, d_recs, n_prob, d_loop, d_sorted, d_cyc, reps)
        if (v == 0) MB_LAUNCH(0);
        else MB_LAUNCH(1);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 3; }
      }
      std::vector<unsigned long long> c(blocks * 2);
      (void)hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost);
      std::vector<uint32_t> got(recs.size());
      (void)hipMemcpy(got.data(), d_sorted, got.size() * 4, hipMemcpyDeviceToHost);
      long bad = 0;
      for (int p0 = 0; p0 < NPROB; ++p0)
        for (int i = 0; i < MB_N; ++i) bad += got[(size_t)p0 * MB_N + i] != want[(size_t)p0 * MB_N + i];
      double loop = 0, cnt = 0;
      for (int b = 0; b < blocks; ++b) { loop += (double)c[2 * b]; cnt += (double)c[2 * b + 1]; }
      if (v >= 1 && blocks == 512) {
        unsigned long long sb[32];
        (void)hipMemcpy(sb, d_cyc + 4096, sizeof sb, hipMemcpyDeviceToHost);
        printf("      wave 0 of block 1, cycles per sort by level: until its tasks are done | barrier wait\n      ");
        for (int i = 0; i < 12; ++i) printf("L%d %.0f|%.0f  ", i, (double)sb[16 + i] / (reps * NPROB), (double)sb[i] / (reps * NPROB));
        printf("\n");
      }
      printf("%4d blocks  %-18s introsort loop %7.0f cycles, counting sort %6.0f   mismatches vs std::sort: %ld\n", blocks, names[v],
             loop / blocks / (reps * NPROB), cnt / blocks / (reps * NPROB), bad);
    }
  return 0;
}
