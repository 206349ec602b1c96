// Diagnostic micro-benchmark (not part of the product): variants of the MaximizeCell greedy scan on one wave, sorted records in LDS.
//   hipcc --offload-arch=gfx950 -O3 -o mb_greedy mb_greedy.hip && ./mb_greedy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

#define R 25
#define S 20
#define N (R * S)
#define NPROB 24

struct Scratch {
  unsigned long long tmpR[64], tmpS[64];
  int left_s[64];
  int quota[64];
  unsigned char owner[64];
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// variant A: the shipped loop
template <bool LOOP>
__device__ __forceinline__ int greedy_a(const uint32_t* s_sorted, Scratch* m, int& got) {
  const int lane = lane_id();
  int left = lane < S ? m->quota[lane] : 0;
  int my_slice = -1;
  unsigned long long free_rbg = (1ull << R) - 1ull;
  unsigned long long open_sl = __ballot(left > 0);
  uint32_t e_next = lane < N ? s_sorted[lane] : 0u;
  for (int c0 = 0; c0 < N && free_rbg != 0ull; c0 += 64) {
    const int i = c0 + lane;
    const uint32_t e = e_next;
    const int rbg = (e >> 8) & 63, sl = e & 63;
    unsigned long long live = __ballot((i < N) & (((free_rbg >> rbg) & (open_sl >> sl) & 1ull) != 0ull));
    int sl_left = __shfl(left, sl, 64);
    asm volatile("" ::: "memory");
    e_next = i + 64 < N ? s_sorted[i + 64] : 0u;
    while (LOOP && live) {
      const int f = __ffsll((long long)live) - 1;
      const int frbg = __builtin_amdgcn_readlane(rbg, f);
      const int fsl = __builtin_amdgcn_readlane(sl, f);
      const int fleft = __builtin_amdgcn_readlane(sl_left, f);
      const unsigned long long same_sl = __ballot(sl == fsl);
      live &= ~__ballot(rbg == frbg);
      free_rbg &= ~(1ull << frbg);
      if (fleft == 1) {
        live &= ~same_sl;
        open_sl &= ~(1ull << fsl);
      }
      if (sl == fsl) sl_left--;
      if (lane == fsl) left--;
      if (lane == frbg) my_slice = fsl;
    }
  }
  if (lane < S) got = m->quota[lane] - left;
  return my_slice;
}

__device__ __forceinline__ unsigned long long rl64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

// variant B: per-chunk same-RBG / same-slice lane masks (LDS atomic OR), scalar-only bookkeeping per grant
template <bool LOOP>
__device__ __forceinline__ int greedy_b(const uint32_t* s_sorted, Scratch* m, int& got) {
  const int lane = lane_id();
  if (lane < S) m->left_s[lane] = m->quota[lane];
  m->owner[lane] = 0xFF;
  m->tmpR[lane] = 0ull;
  m->tmpS[lane] = 0ull;
  int n_taken = 0;
  uint32_t e_next = lane < N ? s_sorted[lane] : 0u;
  const unsigned long long me = 1ull << lane;
  for (int c0 = 0; c0 < N && n_taken < R; c0 += 64) {
    const int i = c0 + lane;
    const uint32_t e = e_next;
    const int rbg = (e >> 8) & 63, sl = e & 63;
    const bool valid = i < N;
    /* masks of the chunk's lanes by RBG and by slice */
    if (valid) {
      atomicOr(&m->tmpR[rbg], me);
      atomicOr(&m->tmpS[sl], me);
    }
    const int own = m->owner[rbg];
    const int sl_left0 = m->left_s[sl];
    const unsigned long long Mr = m->tmpR[rbg], Ms = m->tmpS[sl];
    m->tmpR[lane] = 0ull; /* for the next chunk (LDS operations of one wave execute in order) */
    m->tmpS[lane] = 0ull;
    e_next = i + 64 < N ? s_sorted[i + 64] : 0u;
    unsigned long long live = __ballot(valid && own == 0xFF && sl_left0 > 0);
    unsigned long long taken = 0ull;
    while (LOOP && live) {
      const int f = __ffsll((long long)live) - 1;
      const unsigned long long mr = rl64(Mr, f), ms = rl64(Ms, f);
      const int q = __builtin_amdgcn_readlane(sl_left0, f);
      live &= ~mr;
      const int used = __popcll(taken & ms) + 1;
      if (used == q) live &= ~ms;
      taken |= 1ull << f;
      ++n_taken;
    }
    if ((taken >> lane) & 1ull) {
      m->owner[rbg] = (unsigned char)sl;
      atomicSub(&m->left_s[sl], 1);
    }
  }
  const int o = m->owner[lane];
  if (lane < S) got = m->quota[lane] - m->left_s[lane];
  return o == 0xFF ? -1 : o;
}


// variant E: fixed-point decision of a whole vector of records + compaction of the rest of the stream
// state: free_rbg (scalar), left (lane s)
typedef unsigned int rmask_t; /* R <= 32, S <= 32 */
struct GreedyE {
  rmask_t free_rbg;
  int left, n_taken;
};
__device__ __forceinline__ void process_vec(GreedyE& g, Scratch* m, uint32_t e, bool valid) {
  const int lane = lane_id();
  const unsigned long long me = 1ull << lane, lt = me - 1ull;
  const int rbg = (e >> 8) & 63, sl = e & 63;
  /* lanes of the vector by RBG and by slice (all valid lanes: the ones that are not live never enter T) */
  if (valid) {
    atomicOr(&m->tmpR[rbg], me);
    atomicOr(&m->tmpS[sl], me);
  }
  const int sl_left0 = __shfl(g.left, sl, 64);
  const unsigned long long Mr = m->tmpR[rbg] & lt, Ms = m->tmpS[sl] & lt;
  const unsigned long long Rr = m->tmpR[lane], Rs = m->tmpS[lane];
  asm volatile("" ::: "memory");
  if (valid) {
    m->tmpR[rbg] = 0ull;
    m->tmpS[sl] = 0ull;
  }
  const bool live = valid & (((g.free_rbg >> rbg) & 1u) != 0u) & (sl_left0 > 0);
  const unsigned long long live0 = __ballot(live);
  unsigned long long T = live0;
  for (;;) {
    const bool dup = (Mr & T) != 0ull;
    const int rank = __popcll(Ms & T);
    const unsigned long long Tn = __ballot(!dup & (rank < sl_left0)) & live0;
    if (Tn == T) break;
    T = Tn;
  }
  if ((T >> lane) & 1ull) m->owner[rbg] = (unsigned char)sl;
  g.free_rbg &= ~(rmask_t)__ballot((Rr & T) != 0ull);
  g.left -= __popcll(Rs & T);
  g.n_taken += __popcll(T);
}

__device__ __forceinline__ int greedy_e(uint32_t* s_sorted, Scratch* m, int& got) {
  const int lane = lane_id();
  const unsigned long long lt = (1ull << lane) - 1ull;
  GreedyE g;
  g.free_rbg = (rmask_t)((1ull << R) - 1ull);
  g.left = lane < S ? m->quota[lane] : 0;
  g.n_taken = 0;
  m->owner[lane] = 0xFF;
  int n = N, pos = 0;
  while (g.n_taken < R && pos < n) {
    const int i = pos + lane;
    process_vec(g, m, s_sorted[i < n ? i : 0], i < n);
    pos += 64;
    if (g.n_taken < R && n - pos > 64) {
      /* compact the rest of the stream by what is still live (in place: writes trail the reads) */
      const rmask_t open_sl = (rmask_t)__ballot(g.left > 0);
      int M = 0;
      for (int b = pos; b < n; b += 8 * 64) {
        uint32_t e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int x = b + j * 64 + lane;
          e[j] = s_sorted[x < n ? x : 0];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int x = b + j * 64 + lane;
          const int rbg = (e[j] >> 8) & 63, sl = e[j] & 63;
          const bool live = (x < n) & (((g.free_rbg >> rbg) & (open_sl >> sl) & 1u) != 0u);
          const unsigned long long mk = __ballot(live);
          if (live) s_sorted[M + __popcll(mk & lt)] = e[j];
          M += __popcll(mk);
        }
      }
      n = M;
      pos = 0;
    }
  }
  if (lane < S) got = m->quota[lane] - g.left;
  const int o = m->owner[lane];
  return o == 0xFF ? -1 : o;
}

template <int V>
__global__ void __launch_bounds__(64) bench(const uint32_t* recs, const int* quotas, int* out, unsigned long long* cyc, int reps) {
  __shared__ uint32_t s_sorted[NPROB][N];
  __shared__ Scratch sc;
  if (threadIdx.x < 64) { sc.tmpR[threadIdx.x] = 0; sc.tmpS[threadIdx.x] = 0; }
  __shared__ uint32_t s_work[N];
  __shared__ int s_quota[NPROB][S];
  const int lane = lane_id();
  for (int p = 0; p < NPROB; ++p) {
    for (int i = lane; i < N; i += 64) s_sorted[p][i] = recs[p * N + i];
    if (lane < S) s_quota[p][lane] = quotas[p * S + lane];
  }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  int acc = 0;
  for (int r = 0; r < reps; ++r)
    for (int p = 0; p < NPROB; ++p) {
      if (lane < S) sc.quota[lane] = s_quota[p][lane];
      int got = 0;
      if (V == 4) { for (int i = lane; i < N; i += 64) s_work[i] = s_sorted[p][i]; }
      if (V == 5) { for (int i = lane; i < N; i += 64) s_work[i] = s_sorted[p][i]; acc += s_work[(lane * 7) % N]; continue; }
      const int ms = V == 0 ? greedy_a<true>(s_sorted[p], &sc, got) : V == 1 ? greedy_b<true>(s_sorted[p], &sc, got) : V == 2 ? greedy_a<false>(s_sorted[p], &sc, got) : V == 3 ? greedy_b<false>(s_sorted[p], &sc, got) : greedy_e(s_work, &sc, got);
      acc += ms * 3 + got;
      if (r == 0 && blockIdx.x == 0) {
        if (lane < R) out[p * 64 + lane] = ms;
        if (lane < S) out[p * 64 + 32 + lane] = got;
      }
    }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x7fffffff) out[0] = acc;
}

int main() {
  std::mt19937 g(7);
  std::vector<uint32_t> recs(NPROB * N);
  std::vector<int> quotas(NPROB * S);
  // keys skewed towards the top like MaximizeCell winners
  for (int p = 0; p < NPROB; ++p) {
    std::vector<uint32_t> v;
    for (int r = 0; r < R; ++r)
      for (int s = 0; s < S; ++s) {
        int k = 15 - (int)(std::abs(std::normal_distribution<double>(0, 2.5)(g)));
        if (k < 1) k = 1;
        v.push_back((uint32_t)k << 16 | r << 8 | s);
      }
    std::shuffle(v.begin(), v.end(), g);
    std::stable_sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return (a >> 16) > (b >> 16); });
    std::copy(v.begin(), v.end(), recs.begin() + p * N);
    int q[S];
    for (int s = 0; s < S; ++s) q[s] = 1;
    for (int k = 0; k < R - S; ++k) q[g() % S]++;
    for (int s = 0; s < S; ++s) quotas[p * S + s] = q[s];
  }
  {
    double chunks = 0, depth = 0;
    for (int p = 0; p < NPROB; ++p) {
      int left[S]; bool fr[R]; int nfree = R, last = 0;
      for (int s2 = 0; s2 < S; ++s2) left[s2] = quotas[p * S + s2];
      for (int r = 0; r < R; ++r) fr[r] = true;
      for (int i = 0; i < N && nfree; ++i) {
        uint32_t e = recs[p * N + i]; int r = (e >> 8) & 63, s2 = e & 63;
        if (fr[r] && left[s2] > 0) { fr[r] = false; left[s2]--; nfree--; last = i; }
      }
      depth += last; chunks += last / 64 + 1;
    }
    printf("host: mean depth %.1f, chunks %.2f\n", depth / NPROB, chunks / NPROB);
  }
  uint32_t* d_recs; int* d_q; int* d_out[2]; unsigned long long* d_cyc;
  hipMalloc(&d_recs, recs.size() * 4); hipMalloc(&d_q, quotas.size() * 4);
  hipMalloc(&d_out[0], NPROB * 64 * 4); hipMalloc(&d_out[1], NPROB * 64 * 4); hipMalloc(&d_cyc, 256 * 8); int* d_cyc_out; hipMalloc(&d_cyc_out, NPROB * 64 * 4);
  hipMemcpy(d_recs, recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_q, quotas.data(), quotas.size() * 4, hipMemcpyHostToDevice);
  const int reps = 50;
  std::vector<int> o[2] = {std::vector<int>(NPROB * 64), std::vector<int>(NPROB * 64)};
  for (int v = 0; v < 6; ++v) {
    hipMemset(d_out[v & 1], 0, NPROB * 64 * 4);
    for (int it = 0; it < 2; ++it) {
      if (v == 0) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out[0], d_cyc, reps);
      else if (v == 1) hipLaunchKernelGGL(bench<1>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out[1], d_cyc, reps);
      else if (v == 2) hipLaunchKernelGGL(bench<2>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_cyc_out, d_cyc, reps);
      else if (v == 4) hipLaunchKernelGGL(bench<4>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_out[1], d_cyc, reps);
      else if (v == 5) hipLaunchKernelGGL(bench<5>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_cyc_out, d_cyc, reps);
      else hipLaunchKernelGGL(bench<3>, dim3(256), dim3(64), 0, 0, d_recs, d_q, d_cyc_out, d_cyc, reps);
      hipDeviceSynchronize();
    }
    unsigned long long c[256];
    hipMemcpy(c, d_cyc, sizeof c, hipMemcpyDeviceToHost);
    if (v < 2) hipMemcpy(o[v].data(), d_out[v], NPROB * 64 * 4, hipMemcpyDeviceToHost);
    if (v == 4) { std::vector<int> oe(NPROB * 64); hipMemcpy(oe.data(), d_out[1], NPROB * 64 * 4, hipMemcpyDeviceToHost); int bad = 0; for (size_t i = 0; i < oe.size(); ++i) bad += oe[i] != o[0][i]; printf("   E vs A mismatches: %d\n", bad); }
    printf("variant %c: %.0f cycles per greedy (block 0), %.0f (block 128)\n", 'A' + v, (double)c[0] / (reps * NPROB), (double)c[128] / (reps * NPROB));
  }
  int bad = 0;
  for (size_t i = 0; i < o[0].size(); ++i) bad += o[0][i] != o[1][i];
  printf("mismatches B vs A: %d\n", bad);
  return 0;
}
