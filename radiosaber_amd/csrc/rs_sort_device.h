/*
 * rs_sort_device.h -- the device side of the exact std::sort emulation: the level-synchronous
 * std::__introsort_loop over a whole workgroup (state in LDS for any size, or in registers with EPT array positions per
 * thread) and the stable 16-bucket counting sort that stands for std::__final_insertion_sort.  The serial pieces
 * (median of three, heap-sort fallback) come from rs_sort_emul.h, which the CPU tests run against the real std::sort.
 * Device-only fragment of rs_kernels.hip.
 */
#ifndef RS_SORT_DEVICE_H_
#define RS_SORT_DEVICE_H_

#include "rs_sort_emul.h"
#include "rs_wave.h"

namespace {

/*
 * std::__introsort_loop, level-synchronous over the whole workgroup (the default path).
 * All sub-ranges of one recursion level are partitioned at once, one array position per lane:
 *   M  the lane sitting on a sub-range's first position moves the median of 3 there and publishes
 *      the pivot key (depth 0: it heap-sorts the sub-range instead, as the library does)
 *   F  every position compares with ITS sub-range's pivot; per 64-position chunk two ballots
 *      ("left-scan stop" key <= pivot, "right-scan stop" key >= pivot) go to LDS
 *   R  popcounts over the chunk masks give each stop its rank inside its sub-range:
 *      posA[first + rank from the left] / posB[first + rank from the right]  (= L and Rr)
 *   S  lane j of a sub-range swaps (L[j], Rr[j]) while L[j] < Rr[j]; the lane at the boundary
 *      publishes the cut  L[0] | min(L[k], Rr[k-1])
 *   U  every position moves to its child sub-range [first,cut) or [cut,last); children of at
 *      most 16 elements retire
 * Four workgroup barriers per level, no queue, cost independent of the number of sub-ranges.
 */
/*
 * std::__partial_sort (the heap-sort fallback of std::__introsort_loop when the depth limit runs out) is a serial walk.  It runs
 * on EVERY lane of the wave that owns the sub-range, on wave-uniform operands: the bounds and each element read are scalar values
 * (v_readfirstlane of a same-address LDS read), so the walk is scalar control flow, and every lane stores the same word to the
 * same address.  No divergent single-lane region: round 4 met a long one (SubOpt's hashtable walk under `if (lane == 0)`) that
 * went wrong once the kernel's spill pattern changed (profiles/r05_onelane.md).  RsMisc::heap_sorts counts the calls per site
 * (0 workgroup level of the register form, 1 inside finish_subranges_on_wave, 2 workgroup level of the LDS form) so that a test
 * can tell the branch ran (rs_batch_debug_heap_sorts).
 */
struct UniLdsRef {
  uint32_t* p;
  __device__ __forceinline__ operator uint32_t() const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)*p); }
  __device__ __forceinline__ UniLdsRef& operator=(uint32_t x) { *p = x; return *this; }
  __device__ __forceinline__ UniLdsRef& operator=(const UniLdsRef& o) { *p = (uint32_t)__builtin_amdgcn_readfirstlane((int)*o.p); return *this; }
};
struct UniLdsArr {
  uint32_t* p;
  __device__ __forceinline__ UniLdsRef operator[](int i) const { return UniLdsRef{p + i}; }
};
__device__ __forceinline__ void heap_sort_on_wave(uint32_t* v, int first, int last, Misc* m, int site) {
  UniLdsArr a{v};
  rs_sort::heap_sort(a, __builtin_amdgcn_readfirstlane(first), __builtin_amdgcn_readfirstlane(last));
  if (lane_id() == 0) atomicAdd(&m->heap_sorts[site], 1);
}

__device__ __forceinline__ int count_bits_in(const unsigned long long* masks, int lo, int hi) {
  /* number of set mask bits at positions [lo, hi) */
  if (hi <= lo) return 0;
  const int c0 = lo >> 6, c1 = (hi - 1) >> 6;
  int total = 0;
  for (int c = c0; c <= c1; ++c) {
    unsigned long long mk = masks[c];
    if (c == c0) mk &= ~0ull << (lo & 63);
    if (c == c1) {
      const int h = ((hi - 1) & 63) + 1;
      if (h < 64) mk &= (1ull << h) - 1ull;
    }
    total += __popcll(mk);
  }
  return total;
}

__device__ void introsort_loop_levels(uint32_t* v, int N, uint16_t* posA, uint16_t* posB, uint16_t* segF,
                                      uint16_t* segL, uint16_t* pkbuf, uint16_t* cutbuf, Misc* m) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
  const int n_chunks = (N + 63) >> 6;
  for (int x = tid; x < N; x += nt) {
    segF[x] = 0;
    segL[x] = (uint16_t)(N > 16 ? N : 0);
  }
  if (tid < 48) m->n_level[tid] = (tid == 0 && N > 16) ? 1 : 0;
  if (tid == 0) m->pad[0] = 0; /* entries in the list of sub-ranges to heap-sort */
  int depth = 2 * rs_sort::floor_log2(N > 1 ? N : 1);
  for (int level = 0; level < 47; ++level, --depth) {
    /* M */
    uint32_t* const heap_list = (uint32_t*)posA; /* depth 0: first | last << 16 of the sub-ranges still alive (posA is free here) */
    for (int x = tid; x < N; x += nt) {
      const int L = segL[x];
      if (L != 0 && segF[x] == x) {
        if (depth == 0) {
          heap_list[atomicAdd(&m->pad[0], 1)] = (uint32_t)x | ((uint32_t)L << 16);
        } else {
          LdsArr a{v};
          rs_sort::median_to_first(a, x, x + 1, x + (L - x) / 2, L - 1);
          pkbuf[x] = (uint16_t)(v[x] >> 16);
        }
      }
    }
    __syncthreads();
    if (depth == 0) { /* std::__partial_sort fallback: the waves take the listed sub-ranges in turn (heap_sort_on_wave) */
      const int n_list = m->pad[0];
      for (int j = wave; j < n_list; j += nwaves) {
        const uint32_t ent = heap_list[j];
        heap_sort_on_wave(v, (int)(ent & 0xffffu), (int)(ent >> 16), m, 2);
      }
      __syncthreads();
      break;
    }
    if (m->n_level[level] == 0) break;
    /* F */
    for (int c = wave; c < n_chunks; c += nwaves) {
      const int x = (c << 6) + lane;
      const int L = x < N ? (int)segL[x] : 0;
      const int F = x < N ? (int)segF[x] : 0;
      const int k = x < N ? (int)(v[x] >> 16) : 0;
      const bool in = L != 0 && x > F;
      const int pk = in ? (int)pkbuf[F] : 0;
      const unsigned long long mA = __ballot(in && k <= pk), mB = __ballot(in && k >= pk);
      if (lane == 0) {
        m->maskA[c] = mA;
        m->maskB[c] = mB;
      }
      if (x < N) {
        posA[x] = 0xFFFF;
        posB[x] = 0xFFFF;
      }
    }
    __syncthreads();
    /* R */
    for (int c = wave; c < n_chunks; c += nwaves) {
      const int x = (c << 6) + lane;
      const int L = x < N ? (int)segL[x] : 0;
      const int F = x < N ? (int)segF[x] : 0;
      const int k = x < N ? (int)(v[x] >> 16) : 0;
      const bool in = L != 0 && x > F;
      const int pk = in ? (int)pkbuf[F] : 0;
      if (in && k <= pk) posA[F + count_bits_in(m->maskA, F + 1, x)] = (uint16_t)x;
      if (in && k >= pk) posB[F + count_bits_in(m->maskB, x + 1, L)] = (uint16_t)x;
    }
    __syncthreads();
    /* S */
    for (int x = tid; x < N; x += nt) {
      const int L = segL[x], F = segF[x];
      if (L != 0 && x > F) {
        const int j = x - F - 1;
        const int l = posA[F + j], r = posB[F + j];
        const int l1 = posA[F + j + 1], r1 = posB[F + j + 1]; /* index <= L-1: inside the sub-range */
        const bool sw = l != 0xFFFF && r != 0xFFFF && l < r;
        const bool sw1 = l1 != 0xFFFF && r1 != 0xFFFF && l1 < r1;
        if (sw) {
          const uint32_t a = v[l], b = v[r];
          v[l] = b;
          v[r] = a;
          if (!sw1) cutbuf[F] = (uint16_t)((l1 != 0xFFFF && l1 < r) ? l1 : r);
        } else if (j == 0) {
          cutbuf[F] = (uint16_t)l;
        }
      }
    }
    __syncthreads();
    /* U */
    bool any = false;
    for (int x = tid; x < N; x += nt) {
      const int L = segL[x];
      if (L != 0) {
        const int F = segF[x];
        const int cut = cutbuf[F];
        const int nF = x < cut ? F : cut;
        int nL = x < cut ? cut : L;
        if (nL - nF <= 16) nL = 0;
        segF[x] = (uint16_t)nF;
        segL[x] = (uint16_t)nL;
        any |= nL != 0;
      }
    }
    if (__ballot(any) && lane == 0) atomicAdd(&m->n_level[level + 1], 1);
  }
}

/*
 * The same level-synchronous loop with the per-position state (element, sub-range bounds) held in
 * registers: EPT positions per thread, position x = i*blockDim + tid (so a wave still covers one
 * 64-position chunk per i).  Every lane of a sub-range reads the three median samples itself, so
 * the pivot is known without a publishing step; three workgroup barriers per level.
 */
#if defined(RS_STAMPS) && !defined(RS_STAMPS_HOLD) && !defined(RS_STAMPS_P5)
#define RS_SUBSTAMP(i)                                            \
  do {                                                            \
    if (tid == 0) {                                               \
      unsigned long long now_ = __builtin_readcyclecounter();     \
      sub[i] += now_ - sub_prev;                                  \
      sub_prev = now_;                                            \
    }                                                             \
  } while (0)
#else
#define RS_SUBSTAMP(i) do { } while (0)
#endif



/*
 * One std::__unguarded_partition per live sub-range and level, decided locally from stop counts.
 * In [lo, hi) = (f, l) with pivot key pk, an A-stop is an element with key <= pk (where the upward scan
 * halts), a B-stop one with key >= pk (downward scan).  With A(x) = A-stops in [lo, x) and B(x) = B-stops in
 * (x, hi): the library swaps the j-th A-stop from the left with the j-th B-stop from the right while the
 * former lies left of the latter, so
 *     an A-stop x is swapped  <=>  B(x) > A(x)   (it receives the element of B-stop number A(x) from the right)
 *     a  B-stop x is swapped  <=>  A(x) > B(x)   (it receives the element of A-stop number B(x) from the left)
 * and the returned cut is the leftmost position that is an unswapped A-stop or a swapped B-stop.
 * Swapped elements travel through `xbuf` (A-stop number a at f+a, B-stop number b at l-1-b: they cannot
 * meet, a + b <= len - 3); the cut is an LDS atomicMin per sub-range (slot f>>4: live sub-ranges are longer
 * than 16, so their slots differ).  Counts (round 6): every position writes down how many A-stops of its 64-position chunk lie
 * below it and how many B-stops up to and including it (16 bits, from the two ballots), every chunk its two totals; after the
 * barrier one prefix scan per wave over the totals and, per position, one rank read at each end of its sub-range give
 * A(x) = P_A(x) - P_A(lo), B(x) = P_B(hi - 1) - P_B(x).
 */
/*
 * One wave finishes sub-ranges of at most 64 elements on its own, several at a time when they fit side by side in its 64
 * lanes (lane j holds position fb + j - lb of the sub-range that starts at lane lb and position fb): the same level step as
 * below -- median of three to the front, stop ballots, swap decision from the two stop counts, exchange through `xbuf`, cut =
 * leftmost candidate -- but the pivot candidates come from the other lanes' registers (ds_bpermute), the counts from one
 * ballot pair, and no workgroup barrier is involved: the pieces go down level by level inside the wave until all are at most
 * 16 long.  `l0` = end of my sub-range (0: lane unused), `depth` = introsort's remaining depth at entry (heap-sort fallback
 * when it runs out).  Used for the last, sparsely populated levels: at 500 records they hold 9 / 4 / 2 / 1 sub-ranges of
 * 20-30 elements on average.
 */
/* Round 6: the level step as straight-line code -- the four pivot candidates in ONE batch of register reads (the fourth used to
 * follow the median: a second trip), the median as five selects, the stop ballots as one v_cmp each (keys as floats, NaN outside a
 * piece), the two stop counts from two per-lane 64-bit masks built by one shift each (pieces end at lane 64 at most: ~0 >> (64 - pl)
 * needs no special case), 32-bit counts, and the cut as the first candidate at or above the piece's first inner lane (a piece's
 * candidates are exactly its lanes from the cut on, so no upper bound is needed).  ~95 instructions per level against ~150
 * (profiles/r06_sort_staged.md; the form of rounds 3-5 is in the history up to commit 5593943, behind -DRS_FINISH_V1). */
__device__ __forceinline__ void finish_subranges_on_wave(uint32_t* v, uint32_t* xbuf, int fb, int lb, int l0, int depth, Misc* m) {
  const int lane = lane_id();
  const bool mine = l0 != 0;
  const int x = fb + lane - lb, shift = lb - fb; /* lane = position + shift */
  const unsigned long long lt_lane = (1ull << lane) - 1ull, gt_lane = lane == 63 ? 0ull : (~0ull << (lane + 1));
  uint32_t e = mine ? v[x] : 0u;
  int F = fb, L = l0; /* my piece; L == 0: retired */
  while (__builtin_amdgcn_ballot_w64(L != 0) != 0ull) {
    const bool active = L != 0;
    if (depth == 0) { /* std::__partial_sort fallback: one piece after the other, each on every lane (heap_sort_on_wave) */
      if (mine) v[x] = e;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      unsigned long long lead = __builtin_amdgcn_ballot_w64(active && x == F);
      while (lead != 0ull) {
        const int j = __ffsll((long long)lead) - 1;
        lead &= lead - 1ull;
        heap_sort_on_wave(v, __builtin_amdgcn_readlane(F, j), __builtin_amdgcn_readlane(L, j), m, 1);
      }
      return;
    }
    /* piece bounds in lane space; every lane takes part in the four register reads (a masked-off source lane would
     * read as 0) */
    const int pf = active ? F + shift : 0, pl = active ? L + shift : 2;
    const int ja = pf + 1, jb = pf + (int)((unsigned)(pl - pf) >> 1), jc = pl - 1;
    uint32_t s0 = (uint32_t)__builtin_amdgcn_ds_bpermute(pf << 2, (int)e);
    uint32_t sa = (uint32_t)__builtin_amdgcn_ds_bpermute(ja << 2, (int)e);
    uint32_t sb = (uint32_t)__builtin_amdgcn_ds_bpermute(jb << 2, (int)e);
    uint32_t sc = (uint32_t)__builtin_amdgcn_ds_bpermute(jc << 2, (int)e);
    asm volatile("" : "+v"(s0), "+v"(sa), "+v"(sb), "+v"(sc)); /* all four in one batch */
    const bool ab = (sa >> 16) > (sb >> 16), bc = (sb >> 16) > (sc >> 16), ac = (sa >> 16) > (sc >> 16);
    const uint32_t t1 = ac ? sc : sa, t2 = bc ? sc : sb;
    const int q1 = ac ? jc : ja, q2 = bc ? jc : jb;
    const uint32_t r1 = bc ? sb : t1, r2 = ac ? sa : t2;
    const int u1 = bc ? jb : q1, u2 = ac ? ja : q2;
    const uint32_t sp = ab ? r1 : r2;
    const int pick = ab ? u1 : u2;
    const bool at_f = active & (lane == pf), at_pick = active & (lane == pick);
    e = at_f ? sp : at_pick ? s0 : e;
    const bool in = active & (lane > pf);
    const float kf = in ? (float)(e >> 16) : __builtin_nanf(""), pkf = (float)(sp >> 16);
    const bool isA = kf <= pkf, isB = kf >= pkf;
    const unsigned long long mA = __builtin_amdgcn_ballot_w64(isA), mB = __builtin_amdgcn_ballot_w64(isB);
    const unsigned long long ge_ja = ~0ull << ja, lt_pl = ~0ull >> (64 - pl); /* 1 <= ja <= 63, 2 <= pl <= 64 */
    /* A-stops of my piece left of me, B-stops of my piece right of me */
    const unsigned long long wa = mA & ge_ja & lt_lane, wb = mB & lt_pl & gt_lane;
    const int a = __builtin_popcount((uint32_t)wa) + __builtin_popcount((uint32_t)(wa >> 32));
    const int b = __builtin_popcount((uint32_t)wb) + __builtin_popcount((uint32_t)(wb >> 32));
    const bool swA = isA & (b > a), swB = isB & (a > b);
    const bool sw = swA | swB;
    const int slot = swA ? F + a : L - 1 - b;
    if (sw) xbuf[slot] = e;
    /* candidates = A-stops that stay + B-stops that receive: the two comparisons as wave masks, combined on the scalar side */
    const unsigned long long gBA = __builtin_amdgcn_ballot_w64(b > a), gAB = __builtin_amdgcn_ballot_w64(a > b);
    const unsigned long long mC = ((mA & ~gBA) | (mB & gAB)) & ge_ja;
    const uint32_t got = xbuf[sw ? F + L - 1 - slot : 0];
    e = sw ? got : e;
    /* the first candidate at or above my piece's first inner lane is my piece's cut */
    const int cut_lane = __builtin_ctzll(mC | 0x8000000000000000ull);
    const int cut = cut_lane - shift;
    const bool left = x < cut;
    const int nF = left ? F : cut, nL = left ? cut : L;
    F = active ? nF : F;
    L = (active & (nL - nF > 16)) ? nL : 0;
    --depth;
  }
  if (mine) v[x] = e;
}

template <int EPT, int NT = 0> /* NT: the workgroup's size where it is a compile-time constant (shape-specialised builds), 0: blockDim.x */
__device__ __forceinline__ void introsort_levels_reg(uint32_t* v, int N, uint32_t* xbuf, int32_t* cuts, Misc* m,
                                     unsigned long long* sub, int seg_len = 0, uint16_t* ranks = nullptr) {
  /* sub-ranges per wave at which the last levels go to single waves (same-box A/B, 512 cells: one position per lane -- 500 records --
   * 33.38 M TTIs/s with 2 against 33.18 with 4 and 32.70 with 1; three positions per lane -- 1 280 records -- 13.24 with 2 against 13.41
   * with 4; round 4, 1 280 records: 2 / 4 / 8 within 0.3 % of each other) */
#ifndef RS_FINISH_MAX_1
#define RS_FINISH_MAX_1 2
#endif
#ifndef RS_FINISH_MAX_N
#define RS_FINISH_MAX_N 4
#endif
  constexpr int kFinishMax = EPT == 1 ? RS_FINISH_MAX_1 : RS_FINISH_MAX_N;
  const int tid = threadIdx.x, nt = NT > 0 ? NT : (int)blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
#ifdef RS_STAMPS
  unsigned long long sub_prev = __builtin_readcyclecounter();
#endif
  const int n_chunks = (N + 63) >> 6;
  const unsigned long long lt_lane = (1ull << lane) - 1ull, le_lane = ~0ull >> (63 - lane);
  uint32_t e[EPT];
  int F[EPT], L[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) {
    const int x = i * nt + tid;
    e[i] = x < N ? v[x] : 0u;
    if (seg_len == 0) {
      F[i] = 0;
      L[i] = (x < N && N > 16) ? N : 0;
    } else {
      /* seg_len > 0: the array is a row of independent std::sort calls, seg_len elements each (UpperBound) */
      F[i] = idiv_small(x, seg_len) * seg_len;
      L[i] = (x < N && seg_len > 16) ? F[i] + seg_len : 0;
    }
  }
  const int n_first = seg_len == 0 ? N : seg_len; /* length every std::sort call starts from */
  /* n_level[k]: sub-ranges alive at level k, plus 65536 for each one longer than 64 */
  const int n_calls = seg_len == 0 ? 1 : idiv_small(N, seg_len);
  if (tid < 48) m->n_level[tid] = (tid == 0 && n_first > 16) ? n_calls + (n_first > 64 ? n_calls << 16 : 0) : 0;
  if (tid == 0) m->pad[0] = 0; /* entries in the list of sub-ranges handed to single waves */
  for (int j = tid; j < 128; j += nt) ((uint32_t*)m->maskA)[j] = 0u; /* the two chunk-prefix tables of the levels */
  int depth = 2 * rs_sort::floor_log2(n_first > 1 ? n_first : 1);
  __syncthreads();
  /* Round 6: a wave whose LAST position slot holds no array position at all (1 280 records on 512 threads: 20 chunks, the third slot
   * of waves 4-7 is behind the array's end) skips that slot's instructions in every stage -- wave-uniform, fixed for the whole sort;
   * the slots before it keep running side by side.  (Round 4's skip of every retired chunk lost 3.5 %: it put a branch around each
   * of the three slots on the critical waves.)  -DRS_SORT_NO_LAST_SKIP: off. */
#ifndef RS_SORT_NO_LAST_SKIP
  const bool has_last = EPT == 1 || (EPT - 1) * nwaves + wave < n_chunks;
#else
  const bool has_last = true;
#endif
#define RS_HAS(i) (EPT == 1 || (i) < EPT - 1 || has_last)
  int n_alive = 0;
  for (int level = 0; level < 47; ++level, --depth) {
    /* F, stage 1 (see below): the median-of-3 samples of every position's sub-range, asked for together with the level's count of
     * live sub-ranges -- one LDS round trip at the top of a level instead of two (the reads are harmless whatever the count says) */
    uint32_t s0[EPT], sa[EPT], sb[EPT], sc[EPT];
    int ibx[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      if (!RS_HAS(i)) continue;
      const bool active = L[i] != 0;
      const int f = active ? F[i] : 0, l = active ? L[i] : 2;
      ibx[i] = f + (int)((unsigned)(l - f) >> 1); /* l > f: the halving needs no sign fix */
      s0[i] = v[f];
      sa[i] = v[f + 1];
      sb[i] = v[ibx[i]];
      sc[i] = v[l - 1];
    }
    n_alive = m->n_level[level]; /* complete: the previous level ended with a barrier */
#pragma unroll
    for (int i = 0; i < EPT; ++i)
      if (RS_HAS(i)) asm volatile("" : "+v"(s0[i]), "+v"(sa[i]), "+v"(sb[i]), "+v"(sc[i])); /* keep the reads in this batch */
    /* one test per level; what happens at the two early ends is written behind the loop (round 6: with the ends inside the loop
     * the compiler carried a copy of every position's bounds to each of them, ~20 instructions per level)
     * (a first level that fills every lane -- UpperBound's row of short vectors at one position per lane -- stays a workgroup level) */
    const bool hand_off = depth != 0 && n_alive <= kFinishMax * nwaves && (level > 0 || EPT > 1 || n_alive == 1);
    if (n_alive == 0 || hand_off || depth == 0) break;
    /*
     * Round 6: the three phases in STAGES over the EPT positions of a lane, without a branch.  The form of rounds 2-5 (in the
     * history up to commit 5593943, behind -DRS_SORT_BRANCHY) wrapped every position's work in `if (active)`: the compiler turned that into one block per position
     * with its own s_cbranch_execz and its own s_waitcnt, so the EPT independent chains of a lane ran one after the other --
     * three LDS round trips per phase at three positions per lane, every instruction waiting for the one before it (8.5 cycles
     * each against 5.3 for independent ones, profiles/r03_sort_experiments.md).  Here every stage first issues the LDS reads of
     * ALL positions (a retired position reads the harmless sub-range [0, 2)), then does the arithmetic of all of them with
     * selects; only stores and the atomic stay under a lane mask.  profiles/r06_sort_staged.md.
     */
    /* F: pivot of my sub-range (median of 3, std::__move_median_to_first), stop ballots */
    int rk[EPT]; /* my stop ranks inside my chunk: A-stops below me | B-stops up to me << 8 */
    bool isA[EPT], isB[EPT], moved[EPT];
    /* chunk prefixes (round 6, last step): pre[j] = stops in the chunks before j, A-stops | B-stops << 16.  Every wave ADDS its chunk's
     * two totals to the entries of all later chunks -- one LDS atomic instruction, lanes = the later chunks -- so that R reads its
     * prefixes ready-made beside its two rank reads: ONE LDS round trip where it was a read of the totals, a prefix scan over the
     * lanes and two lane gathers of it.  Two tables by level parity: wave 0 clears the next level's while this one is filled. */
    uint32_t* const pre_w = (uint32_t*)m->maskA + ((level & 1) << 6);
    if (wave == 0) ((uint32_t*)m->maskA)[(((level + 1) & 1) << 6) + lane] = 0u;
    {
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        if (!RS_HAS(i)) { isA[i] = isB[i] = moved[i] = false; rk[i] = 0; continue; }
        const int x = i * nt + tid;
        const int c = i * nwaves + wave;
        const bool active = L[i] != 0;
        /* std::__move_median_to_first with rs_sort::before() = "key greater": five selects on (element, position) each */
        const bool ab = (sa[i] >> 16) > (sb[i] >> 16), bc = (sb[i] >> 16) > (sc[i] >> 16), ac = (sa[i] >> 16) > (sc[i] >> 16);
        const int pa = F[i] + 1, pb = ibx[i], pc_ = L[i] - 1;
        const uint32_t t1 = ac ? sc[i] : sa[i], t2 = bc ? sc[i] : sb[i];
        const int q1 = ac ? pc_ : pa, q2 = bc ? pc_ : pb;
        const uint32_t r1 = bc ? sb[i] : t1, r2 = ac ? sa[i] : t2;
        const int u1 = bc ? pb : q1, u2 = ac ? pa : q2;
        const uint32_t sp = ab ? r1 : r2;
        const int pick = ab ? u1 : u2;
        const bool at_f = active & (x == F[i]), at_pick = active & (x == pick); /* pick > f: never both */
        e[i] = at_f ? sp : at_pick ? s0[i] : e[i];
        moved[i] = at_f | at_pick;
        {
          /* (the store's predicate is computed from a copy the compiler cannot see through: with `if (at_f)` it threaded the
           * selects above over the store's branch and split the wave into "first position" and "others" for a dozen instructions) */
          int fo = F[i];
          asm volatile("" : "+v"(fo));
          if (active & (x == fo)) cuts[fo >> 4] = 0x7fffffff;
        }
        /* keys are 0..15: compared as floats, with NaN for a position outside (f, l), each stop ballot is ONE v_cmp (both
         * comparisons are false on NaN; an integer form needs the range test ANDed in and the mask rebuilt under EXEC) */
        const bool in = active & (x > F[i]);
        const float kf = in ? (float)(e[i] >> 16) : __builtin_nanf("");
        const float pkf = (float)(sp >> 16);
        isA[i] = kf <= pkf;
        isB[i] = kf >= pkf;
        const unsigned long long mA = __builtin_amdgcn_ballot_w64(isA[i]), mB = __builtin_amdgcn_ballot_w64(isB[i]);
        /* Round 6: what a position will be asked for is "how many stops of your chunk lie below (at or below) you": it writes
         * that down -- A-stops below it in the low byte, B-stops up to and including it in the high byte -- and the chunk its
         * two totals; R then needs ONE 16-bit read at its sub-range's first inner position and one at its last instead of two
         * 64-bit masks, two shifted masks and eight mask / count instructions. */
        const int rA = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u));
        const int rB = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u)) + (isB[i] ? 1 : 0);
        rk[i] = rA | (rB << 8);
        if (c < n_chunks) ranks[x] = (uint16_t)rk[i];
        if (lane > c && lane < n_chunks) atomicAdd(&pre_w[lane], (uint32_t)__popcll(mA) | ((uint32_t)__popcll(mB) << 16));
      }
    }
    RS_SUBSTAMP(0);
    __syncthreads();
    RS_SUBSTAMP(1);
    /* R: stop counts -> swap decision; swapped elements to the exchange buffer, cut candidates to the slot */
    int slot[EPT]; /* where my element went / where its replacement arrives; -1: not swapped */
    {
      int plo[EPT], phi[EPT], pcs[EPT], rlo[EPT], rhm[EPT], hm_[EPT];
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        if (!RS_HAS(i)) continue;
        const int x = i * nt + tid;
        if (moved[i]) v[x] = e[i];
        const int lo = F[i] + 1;
        hm_[i] = L[i] != 0 ? L[i] - 1 : 0;
        plo[i] = (int)pre_w[lo >> 6];
        phi[i] = (int)pre_w[hm_[i] >> 6];
        pcs[i] = (int)pre_w[(i * nwaves + wave) & 63];
        rlo[i] = ranks[lo];
        rhm[i] = ranks[hm_[i]];
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        if (!RS_HAS(i)) { slot[i] = -1; continue; }
        const int x = i * nt + tid;
        const int hm = hm_[i];
        const int pc = pcs[i];
        /* A-stops of my sub-range left of me = (stops before me) - (stops before its first inner position); B-stops right of me =
         * (stops up to its last position) - (stops up to me) */
        const int a = ((pc & 0xffff) + (rk[i] & 0xff)) - ((plo[i] & 0xffff) + (rlo[i] & 0xff));
        const int b = ((int)((unsigned)phi[i] >> 16) + (rhm[i] >> 8)) - ((int)((unsigned)pc >> 16) + (rk[i] >> 8));
        const bool swA = isA[i] & (b > a), swB = isB[i] & (a > b); /* never both: b > a excludes a > b */
        const bool sw = swA | swB;
        const int to = swA ? F[i] + a : hm - b;
        slot[i] = sw ? to : -1;
        if (sw) xbuf[to] = e[i];
        const bool cand = (isA[i] & !swA) | swB;
        /* The candidates of a sub-range are exactly its positions from the cut on (right of the cut every position ends with a key
         * that is not before the pivot: an A-stop that stayed or a B-stop that received one; left of it there is none), so the cut is
         * the one candidate whose left neighbour is not one -- the neighbour of a sub-range's first inner position is its pivot
         * slot, never a candidate.  Lane 0 does not see its neighbour and reports too: the slot takes the minimum. */
        const int prev = __builtin_amdgcn_update_dpp(0, cand ? 1 : 0, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const bool report = cand & (prev == 0);
        if (report) atomicMin(&cuts[F[i] >> 4], x);
      }
    }
    RS_SUBSTAMP(2);
    __syncthreads();
    RS_SUBSTAMP(3);
    /* S: receive the swapped element, then move to the child sub-range; sub-ranges of at most 16 retire */
    int alive = 0; /* wave-uniform: sub-ranges of the next level that start in my chunks (+ 65536 per one longer than 64) */
    {
      int cut_[EPT];
      uint32_t rx[EPT];
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        if (!RS_HAS(i)) continue;
        cut_[i] = cuts[F[i] >> 4];
        /* an A-stop's slot f+a pairs with B-stop slot l-1-a and vice versa */
        rx[i] = xbuf[slot[i] >= 0 ? F[i] + L[i] - 1 - slot[i] : 0];
      }
#pragma unroll
      for (int i = 0; i < EPT; ++i) {
        if (!RS_HAS(i)) continue;
        const int x = i * nt + tid;
        const bool live = L[i] != 0, got = slot[i] >= 0;
        e[i] = got ? rx[i] : e[i];
        if (got) v[x] = e[i];
        const bool left = x < cut_[i];
        const int nF = left ? F[i] : cut_[i], nL = left ? cut_[i] : L[i];
        F[i] = live ? nF : F[i];
        L[i] = (live & (nL - nF > 16)) ? nL : 0;
        const int len = x == F[i] ? L[i] - F[i] : 0; /* > 0 on the first position of a live sub-range only (a retired one has L = 0) */
        alive += __popcll(__builtin_amdgcn_ballot_w64(len > 0)) + (__popcll(__builtin_amdgcn_ballot_w64(len > 64)) << 16);
      }
    }
    if (alive != 0 && lane == 0) atomicAdd(&m->n_level[level + 1], alive);
    RS_SUBSTAMP(4);
    __syncthreads();
    RS_SUBSTAMP(5);
#ifdef RS_STAMPS
    if (tid == 0) sub[7] += 1;
#endif
  }
  if (n_alive == 0) return; /* every sub-range is at most 16 long */
  if (depth != 0) {
    /* few sub-ranges left, none longer than 64: their first positions publish them, every wave takes its share and
     * finishes them alone (finish_subranges_on_wave) */
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int x = i * nt + tid;
      const bool leader = L[i] != 0 && x == F[i];
      const unsigned long long mL = __ballot(leader);
      if (mL != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&m->pad[0], __popcll(mL));
        base = __builtin_amdgcn_readfirstlane(base);
        if (leader) cuts[base + __popcll(mL & lt_lane)] = F[i] | (L[i] << 16);
      }
    }
    __syncthreads();
    /* (same-box A/B, 512 cells: the zigzag below 33.17 against 32.99 M TTIs/s at 500 records, one position per lane; at 1 280
     * records -- up to 32 entries to rank -- 13.30 against 13.49: list order there) */
    constexpr bool kZigzag = EPT == 1;
    if constexpr (!kZigzag) {
    /* my share: entries wave, wave + nwaves, ... (lane t fetches the t-th of them), packed side by side while they fit */
    int n_mine = 0; /* entries j = wave + t * nwaves below n_alive (no division: nwaves is a run-time value) */
#pragma unroll
    for (int t = 0; t < kFinishMax; ++t) n_mine += wave + t * nwaves < n_alive ? 1 : 0;
    const int my_ent = lane < n_mine ? cuts[wave + lane * nwaves] : 0;
    for (int t = 0; t < n_mine;) {
      int fb = 0, lb = 0, l0 = 0, used = 0;
      do {
        const int ent = __builtin_amdgcn_readlane(my_ent, t);
        const int f = ent & 0xffff, l = ent >> 16;
        if (used + (l - f) > 64) break;
        if (lane >= used && lane < used + (l - f)) { fb = f; lb = used; l0 = l; }
        used += l - f;
        ++t;
      } while (t < n_mine);
      finish_subranges_on_wave(v, xbuf, fb, lb, l0, depth, m);
    }
    } else {
    /* Every wave reads the whole list (at most kFinishMax * nwaves <= 32 entries, lane j = entry j), ranks the entries
     * by length and takes them in a zigzag over the waves (rank 0..nwaves-1 -> wave 0..nwaves-1, the next nwaves backwards, ...):
     * a wave with two entries gets a long and a short one, which usually fit its 64 lanes together -- entries that do not fit
     * are finished one after the other, and with the entries dealt out in list order one wave of the eight did that in about
     * half of the sorts (two of ~10 sub-ranges of 20-60 elements each on one wave).  Any assignment is correct: the
     * sub-ranges are disjoint. */
    const int ent_l = lane < n_alive ? cuts[lane] : 0;
    const int len_l = lane < n_alive ? (ent_l >> 16) - (ent_l & 0xffff) : -1;
    int rank = 0;
    if constexpr (NT > 0 && kFinishMax * (NT / 64) <= 16) {
      /* round 6: at most 16 entries = one DPP row: fifteen row rotations of a key that holds the length and, below it, the list
       * position reversed (unique keys; an unused lane's is 0) -- two or three instructions per rotation instead of a ten-instruction
       * scalar loop round per entry on every wave */
      const int key = lane < n_alive ? (len_l << 4) | (15 - lane) : 0;
#define RS_ROR_STEP(k) rank += __builtin_amdgcn_update_dpp(0, key, 0x120 + (k) /* row_ror:k */, 0xf, 0xf, false) > key ? 1 : 0
      RS_ROR_STEP(1); RS_ROR_STEP(2); RS_ROR_STEP(3); RS_ROR_STEP(4); RS_ROR_STEP(5);
      RS_ROR_STEP(6); RS_ROR_STEP(7); RS_ROR_STEP(8); RS_ROR_STEP(9); RS_ROR_STEP(10);
      RS_ROR_STEP(11); RS_ROR_STEP(12); RS_ROR_STEP(13); RS_ROR_STEP(14); RS_ROR_STEP(15);
#undef RS_ROR_STEP
    } else {
      for (int j = 0; j < n_alive; ++j) {
        const int lj = __builtin_amdgcn_readlane(len_l, j);
        rank += (lj > len_l || (lj == len_l && j < lane)) ? 1 : 0;
      }
    }
    int blk = 0, posn = rank; /* rank = blk * nwaves + posn without a division (nwaves is a run-time value) */
#pragma unroll
    for (int q = 1; q < kFinishMax; ++q)
      if (rank >= q * nwaves) { blk = q; posn = rank - q * nwaves; }
    const int to_wave = (blk & 1) ? nwaves - 1 - posn : posn;
    unsigned long long mine = __ballot(lane < n_alive && to_wave == wave);
    while (mine != 0ull) {
      int fb = 0, lb = 0, l0 = 0, used = 0;
      unsigned long long rest = mine;
      while (rest != 0ull) { /* pack what fits side by side (in list order; whatever does not fit waits for the next call) */
        const int j = __ffsll((long long)rest) - 1;
        rest &= rest - 1ull;
        const int ent = __builtin_amdgcn_readlane(ent_l, j);
        const int f = ent & 0xffff, l = ent >> 16;
        if (used + (l - f) > 64) continue;
        if (lane >= used && lane < used + (l - f)) { fb = f; lb = used; l0 = l; }
        used += l - f;
        mine &= ~(1ull << j);
      }
      finish_subranges_on_wave(v, xbuf, fb, lb, l0, depth, m);
    }
    }
    __syncthreads();
  } else {
    /* std::__partial_sort fallback for every sub-range still longer than 16: their first positions publish them (the list of
     * the hand-off above; `v` is current, every level writes what it moves), the waves take them in turn (heap_sort_on_wave) */
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int x = i * nt + tid;
      const bool leader = L[i] != 0 && x == F[i];
      const unsigned long long mL = __ballot(leader);
      if (mL != 0ull) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&m->pad[0], __popcll(mL));
        base = __builtin_amdgcn_readfirstlane(base);
        if (leader) cuts[base + __popcll(mL & lt_lane)] = F[i] | (L[i] << 16);
      }
    }
    __syncthreads();
    const int n_list = n_alive & 0xffff;
    for (int j = wave; j < n_list; j += nwaves) {
      const int ent = cuts[j];
      heap_sort_on_wave(v, ent & 0xffff, ent >> 16, m, 0);
    }
    __syncthreads();
  }
}

/*
 * std::__final_insertion_sort == stable sort of the array the introsort loop leaves (a stable order
 * is unique): 16-bucket stable counting sort by DESCENDING key, all waves.
 *   A  every wave, for its 64-element chunks: per-key counts (lane q holds key q) -> hist[chunk][q]
 *   B  wave 0: hist[chunk][q] <- first output slot of key q in that chunk
 *   C  every wave: slot = hist[chunk][key] + (same-key lanes below me); scatter to `out`
 */
__device__ void counting_sort_desc(const uint32_t* v, uint32_t* out, int N, Misc* m) {
  const int lane = lane_id(), wave = wave_id(), nwaves = blockDim.x >> 6;
  const int n_chunks = (N + 63) >> 6;
  for (int c = wave; c < n_chunks; c += nwaves) {
    const int i = (c << 6) + lane;
    const int k = i < N ? (int)(v[i] >> 16) : 0;
    BitBallots<4> bb;
    bb.gather(k, i < N);
    if (lane < 16) m->hist[c * 16 + lane] = (uint16_t)__popcll(bb.lanes_with(lane));
  }
  __syncthreads();
  if (wave == 0) {
#ifdef RS_COUNTING_SORT_V1
    int total = 0;
    if (lane < 16)
      for (int c = 0; c < n_chunks; ++c) total += m->hist[c * 16 + lane];
    int run = 0, acc = 0;
#pragma unroll
    for (int q = 15; q >= 0; --q) {
      int tq = __builtin_amdgcn_readlane(total, q);
      if (lane == q) run = acc;
      acc += tq;
    }
    if (lane < 16)
      for (int c = 0; c < n_chunks; ++c) {
        int h = m->hist[c * 16 + lane];
        m->hist[c * 16 + lane] = (uint16_t)run;
        run += h;
      }
#else
    /* lanes = chunks (at most 64: 4 096 records): per key one prefix scan over the chunks instead of a serial walk of two loops
     * over them on 16 lanes (round 5) */
    int excl[16], base = 0;
#pragma unroll
    for (int q = 15; q >= 0; --q) { /* larger keys first */
      const int h = lane < n_chunks ? (int)m->hist[lane * 16 + q] : 0;
      const int incl = wave_scan_incl(h);
      excl[q] = base + incl - h;
      base += __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane < n_chunks) {
#pragma unroll
      for (int q = 0; q < 16; ++q) m->hist[lane * 16 + q] = (uint16_t)excl[q];
    }
#endif
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int c = wave; c < n_chunks; c += nwaves) {
    const int i = (c << 6) + lane;
    const uint32_t e = i < N ? v[i] : 0;
    const int k = (int)(e >> 16);
    BitBallots<4> bb;
    bb.gather(k, i < N);
    const int rank = __popcll(bb.lanes_with(k) & lt);
    if (i < N) out[m->hist[c * 16 + k] + rank] = e;
  }
  __syncthreads();
}

/* The same stable counting sort when every wave owns at most CPW chunks (chunk j*nwaves + wave, like the
 * register introsort): one ballot pass gives both the per-chunk counts and every element's rank among
 * equal keys in its chunk; after one barrier every wave derives the output offsets of its own chunks
 * from the count table (lane q = key q), so there is no single-wave step and no second barrier. */
/* Round 5, from two chunks per wave on (the 64-RBG grid: 1 280 records = 20 chunks, three per wave -- the phase took 8 100-8 900 cycles
 * there against 1 550 at 500 records, profiles/r04_phase_stamps.log):
 *   A  lanes holding my key = valid & ~OR_b (plane_b ^ -bit_b(my key)): four XORs and three ORs per 32-bit half instead of four
 *      64-bit selects; rank = v_mbcnt of it.  The chunk's count of a key is written by the LAST lane holding it (rank + 1) into rows
 *      the wave has just zeroed itself (its LDS operations execute in order) -- no second mask per key.
 *   B  every wave still derives its chunks' offsets itself (no single-wave step, one barrier), but on all 64 lanes: lane = key +
 *      16 * group, group g adds up chunks g, g + 4, ...; two lane exchanges fold the four groups (values packed two per word).
 *   C  as before.  -DRS_COUNTING_SORT_V1 keeps the round-4 form. */
__device__ __forceinline__ int xor_lanes_sum(int x) { /* x summed over lanes l, l ^ 16, l ^ 32, l ^ 48 */
  const int lane = lane_id();
  x += __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, x);
  x += __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, x);
  return x;
}

template <int CPW, int NT = 0>
__device__ __forceinline__ void counting_sort_desc_owned_v2(const uint32_t* v, uint32_t* out, int N, Misc* m) {
  static_assert(CPW >= 1 && CPW <= 4, "at most four chunks per wave");
  const int lane = lane_id(), wave = wave_id(), nwaves = (NT > 0 ? NT : (int)blockDim.x) >> 6;
  const int n_chunks = (N + 63) >> 6;
  const unsigned long long gt_lane = lane == 63 ? 0ull : (~0ull << (lane + 1));
  uint32_t* const hist32 = (uint32_t*)m->hist;
  uint32_t e[CPW];
  int rank[CPW];
  const bool has_last = CPW == 1 || (CPW - 1) * nwaves + wave < n_chunks; /* (round 6: a wave without a last chunk skips it, as in the levels) */
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    if (CPW > 1 && j == CPW - 1 && !has_last) { e[j] = 0u; rank[j] = 0; continue; }
    const int c = j * nwaves + wave, i = (c << 6) + lane;
    const bool valid = c < n_chunks && i < N;
    e[j] = valid ? v[i] : 0u;
    const int k = (int)(e[j] >> 16);
    if (lane < 8 && c < n_chunks) hist32[c * 8 + lane] = 0u; /* the chunk's 16 counts */
    uint32_t dl = 0u, dh = 0u; /* lanes whose key differs from mine in some bit, as two 32-bit halves */
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t mine = (uint32_t)(((int)(e[j] << (15 - b))) >> 31); /* all ones when my key has bit b (one v_bfe_i32) */
      const unsigned long long plane = __ballot(mine != 0u);
      dl |= (uint32_t)plane ^ mine;
      dh |= (uint32_t)(plane >> 32) ^ mine;
    }
    const unsigned long long vm = __ballot(valid);
    const uint32_t sl = (uint32_t)vm & ~dl, sh = (uint32_t)(vm >> 32) & ~dh;
    rank[j] = (int)__builtin_amdgcn_mbcnt_hi(sh, __builtin_amdgcn_mbcnt_lo(sl, 0u));
    if (valid && ((sl & (uint32_t)gt_lane) | (sh & (uint32_t)(gt_lane >> 32))) == 0u) m->hist[c * 16 + k] = (uint16_t)(rank[j] + 1);
  }
  __syncthreads();
  const int q = lane & 15, g = lane >> 4;
  int total = 0, below[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) below[j] = 0;
  for (int t = 0; t < (n_chunks + 3) >> 2; ++t) { /* (a wave-uniform trip count: unrolled in a shape-specialised build) */
    const int c = 4 * t + g;
    const int h = c < n_chunks ? (int)m->hist[c * 16 + q] : 0;
    total += h;
#pragma unroll
    for (int j = 0; j < CPW; ++j) below[j] += c < j * nwaves + wave ? h : 0;
  }
  /* fold the four groups: counts are at most 4 096, two per word */
  {
    const int w0 = xor_lanes_sum(total | (below[0] << 16));
    total = w0 & 0xffff;
    below[0] = (int)((unsigned)w0 >> 16);
    if constexpr (CPW >= 2) {
      const int w1 = xor_lanes_sum(below[1] | ((CPW >= 3 ? below[CPW >= 3 ? 2 : 0] : 0) << 16));
      below[1] = w1 & 0xffff;
      if constexpr (CPW >= 3) below[2] = (int)((unsigned)w1 >> 16);
    }
    if constexpr (CPW >= 4) below[3] = xor_lanes_sum(below[3]);
  }
  /* elements with a larger key come first: lane q needs the sum of total over keys > q (prefix inside each row of 16) */
  int inc = total;
  {
    int v_ = inc;
    const int identity = 0;
#define RS_ROW_STEP(ctrl, bmask) v_ = v_ + __builtin_amdgcn_update_dpp(identity, v_, ctrl, 0xf, bmask, false)
    RS_ROW_STEP(0x111, 0xf);
    RS_ROW_STEP(0x112, 0xf);
    RS_ROW_STEP(0x114, 0xe);
    RS_ROW_STEP(0x118, 0xc);
#undef RS_ROW_STEP
    inc = v_;
  }
  const int all = __builtin_amdgcn_readlane(inc, 15);
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    if (CPW > 1 && j == CPW - 1 && !has_last) continue;
    const int c = j * nwaves + wave, i = (c << 6) + lane;
    const int base_q = all - inc + below[j]; /* lane q < 16 */
    const int base = __builtin_amdgcn_ds_bpermute((int)(e[j] >> 16) << 2, base_q); /* every lane: no branch around it */
    if (c < n_chunks && i < N) out[base + rank[j]] = e[j];
  }
  __syncthreads();
}

template <int CPW, int NT = 0>
__device__ __forceinline__ void counting_sort_desc_owned(const uint32_t* v, uint32_t* out, int N, Misc* m) {
#ifndef RS_COUNTING_SORT_V1
#ifndef RS_COUNTING_SORT_V2_MIN_CPW
#define RS_COUNTING_SORT_V2_MIN_CPW 1 /* round 6: the ballot form wins at one chunk per wave too (42.4 against 42.0 M TTIs/s at the headline shape, same lease) */
#endif
  if constexpr (CPW >= RS_COUNTING_SORT_V2_MIN_CPW) {
    counting_sort_desc_owned_v2<CPW, NT>(v, out, N, m);
    return;
  }
#endif
  const int lane = lane_id(), wave = wave_id(), nwaves = (NT > 0 ? NT : (int)blockDim.x) >> 6;
  const int n_chunks = (N + 63) >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t e[CPW];
  int rank[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = j * nwaves + wave, i = (c << 6) + lane;
    const bool valid = c < n_chunks && i < N;
    e[j] = valid ? v[i] : 0u;
    BitBallots<4> bb;
    bb.gather((int)(e[j] >> 16), valid);
    rank[j] = __popcll(bb.lanes_with((int)(e[j] >> 16)) & lt);
    if (lane < 16 && c < n_chunks) m->hist[c * 16 + lane] = (uint16_t)__popcll(bb.lanes_with(lane));
  }
  __syncthreads();
  int total = 0, below[CPW];
#pragma unroll
  for (int j = 0; j < CPW; ++j) below[j] = 0;
  if (lane < 16)
    for (int c = 0; c < n_chunks; ++c) {
      const int h = m->hist[c * 16 + lane];
#pragma unroll
      for (int j = 0; j < CPW; ++j) below[j] += c < j * nwaves + wave ? h : 0;
      total += h;
    }
  /* elements with a larger key come first: lane q needs the sum of total over keys > q */
  int inc = total;
  {
    int v_ = inc;
    const int identity = 0;
#define RS_ROW_STEP(ctrl, bmask) v_ = v_ + __builtin_amdgcn_update_dpp(identity, v_, ctrl, 0xf, bmask, false)
    RS_ROW_STEP(0x111, 0xf);
    RS_ROW_STEP(0x112, 0xf);
    RS_ROW_STEP(0x114, 0xe);
    RS_ROW_STEP(0x118, 0xc);
#undef RS_ROW_STEP
    inc = v_;
  }
  const int all = __builtin_amdgcn_readlane(inc, 15);
#pragma unroll
  for (int j = 0; j < CPW; ++j) {
    const int c = j * nwaves + wave, i = (c << 6) + lane;
    const int base_q = all - inc + below[j]; /* lane q < 16 */
    const int base = __builtin_amdgcn_ds_bpermute((int)(e[j] >> 16) << 2, base_q); /* every lane: no branch around it */
    if (c < n_chunks && i < N) out[base + rank[j]] = e[j];
  }
  __syncthreads();
}

}  // namespace

#endif /* RS_SORT_DEVICE_H_ */
