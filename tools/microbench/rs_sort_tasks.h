/*
 * rs_sort_tasks.h -- EXPERIMENT, not part of the product (round 3; profiles/r03_sort_experiments.md): the introsort loop of
 * rs_sort_device.h as one task per wave.  Bit-exact against std::sort (mb_sort.hip), slower than the level-synchronous
 * workgroup form on MI355X at every size tried, so the product keeps the latter.  The hybrid (workgroup levels that hand over
 * to tasks once every sub-range fits TASK_K registers per lane) needed a hook inside introsort_levels_reg; it lives in commit
 * 7d8dba5 only.
 */
#ifndef RS_SORT_TASKS_H_
#define RS_SORT_TASKS_H_

#include "rs_sort_device.h"

namespace {

/*
 * Task-per-wave form of the introsort loop (round 3).  A task = one sub-range [f, l) of one recursion level; ONE wave partitions
 * it alone with the whole sub-range in its registers, K = ceil(len / 64) elements per lane in chunk-major order (local index
 * q = x - f sits in register q / 64 of lane q % 64):
 *   - the elements and the four median-of-3 samples come from LDS in one batch of reads; the pivot and the sample that goes to
 *     the front are wave-uniform values, the swap is applied in registers;
 *   - per register, two v_cmp write the stop masks (A: key <= pivot, B: key >= pivot; NaN outside (f, l)) straight into scalar
 *     register pairs, s_bcnt1 gives the chunk counts and a scalar running sum the prefixes -- no prefix scan across lanes, no
 *     mask table in LDS, no ds_bpermute: a(x) = v_mbcnt(maskA, A-stops in earlier registers), b'(x) = B-stops at or before x;
 *   - the swap rule of the level-synchronous form (an A-stop moves iff B(x) > A(x), a B-stop iff A(x) > B(x), with
 *     B(x) = totB - b'(x)) becomes a + b' < totB / a + b' > totB; swapped elements cross through `xbuf` exactly as there;
 *   - the cut is the first candidate bit of the first register that has one (s_ff1), known to the wave without an atomic.
 * A level of the recursion is then: every wave takes the tasks t = wave, wave + nwaves, ... of the level's list, partitions
 * them, and appends the children longer than 16 to the next level's list; ONE workgroup barrier per level (the level-
 * synchronous form needs three, and ~240 instructions on EVERY wave whatever the number of sub-ranges).  Waves without a task
 * wait at the barrier and leave their issue slots to the co-resident cell.  At 500 records the recursion is 8.2 levels deep
 * on average with 1, 2, 4, 7.5, 11, 10, 5, 2.5, ... tasks per level (tools/sort_study.py).
 * `tasks`: two lists of RS_TASK_CAP entries (f | l << 16), by level parity.  KMAX * 64 >= the longest sub-range handed in.
 */
#define RS_TASK_CAP 256
/* one task on one wave, K = ceil(len / 64) registers per lane known at compile time (only the last register is partial) */
template <int K>
__device__ __forceinline__ int partition_task_k(uint32_t* v, uint32_t* xbuf, const int f, const int n) {
  const int lane = lane_id();
  uint32_t e[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int q = i * 64 + lane;
    e[i] = (i < K - 1 || q < n) ? v[f + q] : 0u;
  }
  /* std::__move_median_to_first(first, first + 1, first + len / 2, last - 1): same batch of LDS reads, uniform addresses */
  const int qb = (int)((unsigned)n >> 1), qc = n - 1;
  const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[f]);
  const uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[f + 1]);
  const uint32_t sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[f + qb]);
  const uint32_t sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)v[f + qc]);
  int pickq;
  uint32_t sp;
  if (rs_sort::before(sa, sb)) {
    if (rs_sort::before(sb, sc)) { pickq = qb; sp = sb; }
    else if (rs_sort::before(sa, sc)) { pickq = qc; sp = sc; }
    else { pickq = 1; sp = sa; }
  } else if (rs_sort::before(sa, sc)) { pickq = 1; sp = sa; }
  else if (rs_sort::before(sb, sc)) { pickq = qc; sp = sc; }
  else { pickq = qb; sp = sb; }
  if (lane == 0) e[0] = sp;
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (K == 1 || i == (pickq >> 6)) {
      if (lane == (pickq & 63)) e[i] = s0;
    }
  const float pkf = (float)(sp >> 16);
  /* pass 1: stop masks (scalar register pairs), ranks */
  unsigned long long mA[K], mB[K];
  int a[K], bp[K];
  int accA = 0, accB = 0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int q = i * 64 + lane;
    float kf = (float)(e[i] >> 16);
    if (i == 0) kf = lane == 0 ? __builtin_nanf("") : kf;       /* q = 0 holds the pivot: outside the scanned range */
    if (i == K - 1) kf = q < n ? kf : __builtin_nanf("");       /* beyond the sub-range */
    mA[i] = __ballot(kf <= pkf);
    mB[i] = __ballot(kf >= pkf);
    a[i] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mA[i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mA[i], (unsigned)accA));
    bp[i] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mB[i] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mB[i], (unsigned)accB)) +
            (int)((mB[i] >> lane) & 1ull);
    accA += __popcll(mA[i]);
    accB += __popcll(mB[i]);
  }
  /* pass 2: who moves, where to; first candidate = the cut */
  const int totB = accB, cB = n - 1 - totB;
  int slot[K];
  int cut = -1;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int t = a[i] + bp[i];
    const unsigned long long swA = mA[i] & __ballot(t < totB), swB = mB[i] & __ballot(t > totB); /* never both */
    const bool mineA = ((swA >> lane) & 1ull) != 0ull, mine = (((swA | swB) >> lane) & 1ull) != 0ull;
    slot[i] = mineA ? a[i] : cB + bp[i];
    if (mine) xbuf[f + slot[i]] = e[i];
    else slot[i] = -1;
    const unsigned long long mC = (mA[i] & ~swA) | swB;
    if (cut < 0 && mC != 0ull) cut = i * 64 + __ffsll((long long)mC) - 1;
  }
  /* pass 3: receive, write back what changed */
#pragma unroll
  for (int i = 0; i < K; ++i)
    if (slot[i] >= 0) e[i] = xbuf[f + n - 1 - slot[i]];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int q = i * 64 + lane;
    if (slot[i] >= 0 || q == 0 || q == pickq) v[f + q] = e[i];
  }
  return f + cut;
}

template <int KMAX>
__device__ __forceinline__ int partition_task_on_wave(uint32_t* v, uint32_t* xbuf, const int f, const int l) {
  const int n = l - f; /* wave-uniform, 16 < n <= 64 * KMAX */
  const int K = (n + 63) >> 6;
  static_assert(KMAX >= 1 && KMAX <= 8, "dispatch below");
  if (KMAX >= 8 && K >= 8) return partition_task_k<(KMAX >= 8 ? 8 : 1)>(v, xbuf, f, n);
  if (KMAX >= 7 && K == 7) return partition_task_k<(KMAX >= 7 ? 7 : 1)>(v, xbuf, f, n);
  if (KMAX >= 6 && K == 6) return partition_task_k<(KMAX >= 6 ? 6 : 1)>(v, xbuf, f, n);
  if (KMAX >= 5 && K == 5) return partition_task_k<(KMAX >= 5 ? 5 : 1)>(v, xbuf, f, n);
  if (KMAX >= 4 && K == 4) return partition_task_k<(KMAX >= 4 ? 4 : 1)>(v, xbuf, f, n);
  if (KMAX >= 3 && K == 3) return partition_task_k<(KMAX >= 3 ? 3 : 1)>(v, xbuf, f, n);
  if (KMAX >= 2 && K == 2) return partition_task_k<(KMAX >= 2 ? 2 : 1)>(v, xbuf, f, n);
  return partition_task_k<1>(v, xbuf, f, n);
}

/* the introsort loop from a list of tasks: `level` = their recursion level (list half level & 1; m->n_level[level] = count in the
 * low half, number of tasks longer than 64 in the high half; made visible by a barrier), `depth` = introsort's remaining depth
 * there.  FINISH: once no task is longer than 64 and every wave's share is at most RS_WAVE_FINISH_MAX, the waves finish their
 * shares alone, packed side by side (finish_subranges_on_wave), instead of meeting at a barrier per level.  Ends with a barrier. */
template <int KMAX, bool FINISH>
__device__ __forceinline__ void introsort_task_levels(uint32_t* v, uint32_t* xbuf, int32_t* tasks, Misc* m, int level, int depth,
                                                      unsigned long long* sub = nullptr) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
#ifdef RS_TASK_STAMPS
  unsigned long long sub_prev = __builtin_readcyclecounter();
#endif
  for (; level < 47; ++level, --depth) {
    const int word = m->n_level[level];
    const int n_t = word & 0xffff;
    if (n_t == 0) break;
    const int32_t* cur = tasks + (level & 1) * RS_TASK_CAP;
    int32_t* nxt = tasks + ((level + 1) & 1) * RS_TASK_CAP;
    if (depth == 0) { /* std::__partial_sort fallback for every sub-range still alive */
      for (int t = tid; t < n_t; t += nt) {
        LdsArr arr{v};
        rs_sort::heap_sort(arr, cur[t] & 0xffff, (int)((unsigned)cur[t] >> 16));
      }
      __syncthreads();
      break;
    }
    if (FINISH && (word >> 16) == 0 && n_t <= RS_WAVE_FINISH_MAX * nwaves) {
      /* my share: entries wave, wave + nwaves, ... (lane t fetches the t-th of them), packed side by side while they fit */
      int n_mine = 0;
#pragma unroll
      for (int t = 0; t < RS_WAVE_FINISH_MAX; ++t) n_mine += wave + t * nwaves < n_t ? 1 : 0;
      const int my_ent = lane < n_mine ? cur[wave + lane * nwaves] : 0;
      for (int t = 0; t < n_mine;) {
        int fb = 0, lb = 0, l0 = 0, used = 0;
        do {
          const int ent = __builtin_amdgcn_readlane(my_ent, t);
          const int f = ent & 0xffff, l = (int)((unsigned)ent >> 16);
          if (used + (l - f) > 64) break;
          if (lane >= used && lane < used + (l - f)) { fb = f; lb = used; l0 = l; }
          used += l - f;
          ++t;
        } while (t < n_mine);
        finish_subranges_on_wave(v, xbuf, fb, lb, l0, depth, m);
      }
      __syncthreads();
      break;
    }
    for (int t = __builtin_amdgcn_readfirstlane(wave); t < n_t; t += nwaves) {
      const int ent = __builtin_amdgcn_readfirstlane(cur[t]);
      const int f = ent & 0xffff, l = (int)((unsigned)ent >> 16);
      const int cut = partition_task_on_wave<KMAX>(v, xbuf, f, l);
      const int c0 = cut - f > 16 ? 1 : 0, c1 = l - cut > 16 ? 1 : 0;
      if ((c0 | c1) && lane == 0) {
        const int big = (cut - f > 64 ? 1 : 0) + (l - cut > 64 ? 1 : 0);
        const int base = atomicAdd(&m->n_level[level + 1], c0 + c1 + (big << 16)) & 0xffff;
        if (c0) nxt[base] = f | (cut << 16);
        if (c1) nxt[base + c0] = cut | (l << 16);
      }
    }
#ifdef RS_TASK_STAMPS
    if (tid == 0 && sub) { const unsigned long long now_ = __builtin_readcyclecounter(); sub[16 + (level < 15 ? level : 15)] += now_ - sub_prev; sub_prev = now_; }
#endif
    __syncthreads();
#ifdef RS_TASK_STAMPS
    if (tid == 0 && sub) { const unsigned long long now_ = __builtin_readcyclecounter(); sub[level < 15 ? level : 15] += now_ - sub_prev; sub_prev = now_; }
#endif
  }
}

/* std::__introsort_loop of N elements (or of N / seg_len independent calls of seg_len elements each: UpperBound) in task form;
 * sub-ranges longer than KMAX * 64 first go through workgroup levels (introsort_levels_reg with `task_max`). */
template <int KMAX, int EPT>
__device__ __forceinline__ void introsort_tasks(uint32_t* v, int N, uint32_t* xbuf, int32_t* cuts, Misc* m, int seg_len = 0,
                                                unsigned long long* sub = nullptr) {
  const int tid = threadIdx.x, nt = blockDim.x;
  int32_t* tasks = (int32_t*)m->hist; /* free during the introsort loop: two lists of RS_TASK_CAP entries */
  const int n_first = seg_len == 0 ? N : seg_len;
  const int n_calls = seg_len == 0 ? 1 : idiv_small(N, seg_len);
  if (tid < 48) m->n_level[tid] = (tid == 0 && n_first > 16) ? n_calls + (n_first > 64 ? n_calls << 16 : 0) : 0;
  for (int t = tid; t < n_calls && n_first > 16; t += nt) tasks[t] = (t * n_first) | ((t * n_first + n_first) << 16);
  __syncthreads();
  (void)cuts;
  introsort_task_levels<KMAX, true>(v, xbuf, tasks, m, 0, 2 * rs_sort::floor_log2(n_first > 1 ? n_first : 1), sub);
}

}  // namespace

#endif /* RS_SORT_TASKS_H_ */
