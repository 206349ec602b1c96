/* Inter-slice policies of DownlinkTransportScheduler::RBsAllocation (ref: downlink-transport-scheduler.cpp:570-586) that run on
 * ONE wave once the per-(RBG, slice) winners are known: GreedyByRow (inter_sched_ 0), SubOpt (1), the greedy scan of
 * MaximizeCell (2, over the records rs_sort_device.h sorted) and VogelApproximate (3).  Each returns, in lane r, the slice
 * RBG r goes to (-1: none) and leaves in `got`, lane s, the RBGs granted to slice s; `m->quota[s]` holds the quotas.
 * Records are u32: CQI key of the slice's winner << 16 | rbg << 8 | slice (key 0: the slice has no user).  Efficiencies are
 * strictly increasing in the key, so comparisons use keys and only differences use the doubles (RsMisc::eff16).
 * S_T / R_T: the shape as template constants in a shape-specialised build (0: run-time values).
 * Included by rs_kernels.hip and embedded for the run-time specialisation (rs_jit.cpp). */
#ifndef RS_INTERSLICE_H_
#define RS_INTERSLICE_H_

#include "rs_device.h"
#include "rs_sort_emul.h"
#include "rs_wave.h"

namespace {

#define RS_GBR_GROUP 2 /* RBGs decided per step of GreedyByRow */

template <int S_T, int R_T>
__device__ __forceinline__ int interslice_greedy_by_row(const uint32_t* s_elems, const RsMisc* m, int S_rt, int R_rt, int& got) {
  const int lane = lane_id();
  const int S = S_T ? S_T : S_rt, R = R_T ? R_T : R_rt; /* front-end constants in a shape-specialised build */
  /* GreedyByRow, ref: :249-272 -- RBG ascending, argmax eff over slices under quota, first max
   * wins.  eff is strictly increasing in CQI (0 for an empty slice), so integer keys compare alike. */
  const int quota = lane < S ? m->quota[lane] : 0;
  int my_slice = -1;
  /* RS_GBR_GROUP RBGs per step: their arg-max reductions run side by side (independent DPP chains) under the quotas as they
   * stand at the start of the step.  Slices only ever close, so a result stays the arg-max as long as its slice is still open
   * when its turn comes; otherwise (an earlier RBG of the step took the slice's last unit) it is redone.  The keys of the
   * next step are loaded meanwhile. */
  constexpr int GN = RS_GBR_GROUP;
  const uint32_t* colp = s_elems + (lane < S ? lane : 0);
  auto key_of = [&](int r) -> int { return (int)(colp[(r < R ? r : R - 1) * S] >> 16); };
  auto pick = [&](int key) -> int {
    const bool ok = lane < S && got < quota;
    const int bestp = wave_max(ok ? (key << 6) | (63 - lane) : -1);
    return bestp < 0 ? -1 : 63 - (bestp & 63);
  };
  int kn[GN];
#pragma unroll
  for (int j = 0; j < GN; ++j) kn[j] = key_of(j);
  for (int r = 0; r < R; r += GN) {
    int kc[GN], pj[GN];
    const bool ok = lane < S && got < quota;
#pragma unroll
    for (int j = 0; j < GN; ++j) {
      kc[j] = kn[j];
      kn[j] = key_of(r + GN + j);
    }
#pragma unroll
    for (int j = 0; j < GN; ++j) pj[j] = wave_max(ok ? (kc[j] << 6) | (63 - lane) : -1);
#pragma unroll
    for (int j = 0; j < GN; ++j) {
      if (r + j < R) {
        int sj = pj[j] < 0 ? -1 : 63 - (pj[j] & 63);
        if (j > 0 && sj >= 0 && __builtin_amdgcn_readlane(got, sj) >= __builtin_amdgcn_readlane(quota, sj)) sj = pick(kc[j]);
        if (lane == sj) got++;
        if (lane == r + j) my_slice = sj;
      }
    }
  }
  return my_slice;
}

/* s_sorted: the R*S records in std::sort's order.  One record per step: the form used above 32 RBGs (the vector form below up to 32) */
template <int S_T, int R_T>
__device__ __forceinline__ int interslice_maximize_cell(const uint32_t* s_sorted, const RsMisc* m, int S_rt, int R_rt, int& got
#ifdef RS_STAMPS
                                                        , unsigned long long* stamp_acc
#endif
) {
  const int lane = lane_id();
  const int S = S_T ? S_T : S_rt, R = R_T ? R_T : R_rt; /* front-end constants in a shape-specialised build */
#ifdef RS_STAMPS
  const int tid = threadIdx.x;
#endif
  /* MaximizeCell greedy scan, ref: :362-369: sorted records in order, take the RBG if it is
   * free and the slice is under quota */
  const int N = R * S;
  int left = lane < S ? m->quota[lane] : 0; /* lane s: quota[s] - granted[s] */
  int my_slice = -1;                        /* lane r: slice that got RBG r */
#ifdef RS_STAMPS
  int assigned = 0;
  int scan_end = 0; /* diagnostic: how deep the scan went */
#endif
  /* free RBGs and slices still under quota as two scalar masks: a record is live when both of its bits are set (a 64-bit
   * shift per lane instead of two cross-lane reads); the next 64 records are fetched while this chunk is scanned.  Every
   * record lane carries the remaining quota of its own slice, so the three facts about the first live record (RBG, slice,
   * is this the slice's last RBG) are three independent lane reads */
  unsigned long long free_rbg = R >= 64 ? ~0ull : (1ull << R) - 1ull;
  unsigned long long open_sl = __ballot(left > 0);
  uint32_t e_next = lane < N ? s_sorted[lane] : 0u;
  for (int c0 = 0; c0 < N && free_rbg != 0ull; c0 += 64) {
    const int i = c0 + lane;
    const uint32_t e = e_next;
    const int rbg = (e >> 8) & 63, sl = e & 63;
    /* records of this chunk that can still be taken: RBG free and slice under quota */
    unsigned long long live = __ballot((i < N) & (((free_rbg >> rbg) & (open_sl >> sl) & 1ull) != 0ull));
    int sl_left = __shfl(left, sl, 64); /* quota left of my record's slice (all lanes take part) */
    asm volatile("" ::: "memory"); /* the fetch of the next chunk is issued here, not above the wait for this one */
    e_next = i + 64 < N ? s_sorted[i + 64] : 0u;
    while (live) {
      const int f = __ffsll((long long)live) - 1;
      const int frbg = __builtin_amdgcn_readlane(rbg, f);
      const int fsl = __builtin_amdgcn_readlane(sl, f);
      const int fleft = __builtin_amdgcn_readlane(sl_left, f);
      const unsigned long long same_sl = __ballot(sl == fsl);
      live &= ~__ballot(rbg == frbg); /* the RBG is gone (this drops record f too) */
      free_rbg &= ~(1ull << frbg);
      if (fleft == 1) { /* the slice just used its last RBG */
        live &= ~same_sl;
        open_sl &= ~(1ull << fsl);
      }
      if (sl == fsl) sl_left--;
      if (lane == fsl) left--;
      if (lane == frbg) my_slice = fsl;
#ifdef RS_STAMPS
      assigned++;
      scan_end = c0 + f;
#endif
    }
  }
#ifdef RS_STAMPS
  if (tid == 0) { /* how deep the greedy scan went: last sorted position it looked at */
    stamp_acc[9] += (unsigned long long)assigned;
    stamp_acc[10] += (unsigned long long)scan_end;
  }
#endif
  if (lane < S) got = m->quota[lane] - left;
  return my_slice;
}

/*
 * MaximizeCell greedy scan, ref: :362-369 -- the sorted records in order; a record takes its RBG when the RBG is still free and
 * its slice is under quota.  A single wave issues one instruction every ~8 cycles whatever the instruction is, so the
 * record-at-a-time loop (22 instructions per granted RBG + 28 per chunk of 64 records) costs ~7 000 cycles at 25 RBGs.  Here a
 * whole vector of 64 records is decided at once:
 *   taken(i) = live(i) and no taken(k < i) has i's RBG and fewer than left[slice(i)] taken(k < i) have i's slice
 * is a recursion on the position, so the map T -> F(T) (evaluate the right-hand side with T in place of `taken`) has exactly
 * one fixed point, the scan's answer, and iterating it from T = live reaches it: after n rounds the first n positions are final
 * (measured: ~6 rounds for the first 64 records, which hold ~18 of 25 grants, ~4 for the rest).  A round is two masked popcounts
 * per lane -- the lanes of the vector that hold my RBG / my slice and lie before me come from one LDS atomic OR per lane.
 * After the first vector the rest of the array is compacted in place to the records that are still live (~40 of 436), which
 * normally fit one more vector; if not, the compaction repeats after every vector.
 * s_sorted is consumed (overwritten by the compaction); scratch: m->maskA / maskB (lane masks by RBG / by slice) and m->n_level
 * (the slice every RBG went to), all free between the sort and the end of the TTI.
 */
template <bool K32> struct RsMaskT { typedef unsigned long long type; };
template <> struct RsMaskT<true> { typedef uint32_t type; };

/* one vector: records src[pos .. pos+64) of a stream of n, decided on the calling wave; free_rbg / left / n_taken updated */
template <class set_t>
__device__ __forceinline__ void rs_scan_decide_vector(const uint32_t* src, int pos, int n, RsMisc* m, set_t& free_rbg, int& left,
                                                      int& n_taken) {
  const int lane = lane_id();
  const unsigned long long me = 1ull << lane, lt = me - 1ull;
  unsigned long long* const by_rbg = m->maskA;
  unsigned long long* const by_slice = m->maskB;
  unsigned char* const owner_of = (unsigned char*)m->n_level;
  const int i = pos + lane;
  const bool valid = i < n;
  const uint32_t e = src[valid ? i : 0];
  const int rbg = (e >> 8) & 63, sl = e & 63;
  /* lanes of this vector by RBG and by slice (every valid lane: the ones that are not live never enter T) */
  if (valid) {
    atomicOr(&by_rbg[rbg], me);
    atomicOr(&by_slice[sl], me);
  }
  const int sl_left = __shfl(left, sl, 64); /* quota left of my record's slice (all lanes take part) */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned long long before_r = by_rbg[rbg] & lt, before_s = by_slice[sl] & lt;
  const unsigned long long of_rbg = by_rbg[lane], of_slice = by_slice[lane]; /* lane r: records of RBG r; lane s: of slice s */
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (valid) { /* clean for the next vector */
    by_rbg[rbg] = 0ull;
    by_slice[sl] = 0ull;
  }
  /* (round 6: every wave mask below is the AND of ballots of plain comparisons -- a ballot of a combined predicate costs a
   * v_cndmask + v_cmp on top -- and a lane's own bit of T is the predicate it was built from) */
  const bool rbg_free = ((free_rbg >> rbg) & 1) != 0, under = sl_left > 0;
  const unsigned long long live = __builtin_amdgcn_ballot_w64(valid) & __builtin_amdgcn_ballot_w64(rbg_free) & __builtin_amdgcn_ballot_w64(under);
  /* (the iteration converges to the scan's answer from ANY start -- position 0 is final after one round whatever T was, position 1
   * after two, ... -- so it starts from the live records that are the first live one of their RBG in this vector, which is what
   * two or three rounds from T = live arrive at first) */
  const uint32_t dup0 = (uint32_t)(before_r & live) | (uint32_t)((before_r & live) >> 32);
  unsigned long long T = live & __builtin_amdgcn_ballot_w64(dup0 == 0u);
  bool mine = false; /* my bit of T */
  for (;;) {
    const uint32_t gone = (uint32_t)(before_r & T) | (uint32_t)((before_r & T) >> 32);
    const int used = __builtin_popcount((uint32_t)(before_s & T)) + __builtin_popcount((uint32_t)((before_s & T) >> 32));
    const bool keep_r = gone == 0u, keep_s = used < sl_left;
    const unsigned long long Tn = __builtin_amdgcn_ballot_w64(keep_r) & __builtin_amdgcn_ballot_w64(keep_s) & live;
    mine = keep_r & keep_s;
    if (Tn == T) break;
    T = Tn;
  }
  if (mine & valid & rbg_free & under) owner_of[rbg] = (unsigned char)sl;
  free_rbg &= ~(set_t)__ballot((of_rbg & T) != 0ull);
  left -= __popcll(of_slice & T);
  n_taken += __popcll(T);
}

template <int S_T, int R_T, bool K32> /* K32: at most 32 RBGs and 32 slices */
__device__ __forceinline__ int interslice_maximize_cell_vector(uint32_t* s_sorted, RsMisc* m, int S_rt, int R_rt, int& got
#ifdef RS_STAMPS
                                                        , unsigned long long* stamp_acc
#endif
) {
  const int lane = lane_id();
  const int S = S_T ? S_T : S_rt, R = R_T ? R_T : R_rt; /* front-end constants in a shape-specialised build */
  const int N = R * S;
  /* RBG and slice sets as 32-bit scalars when they fit (64-bit shifts by a lane value are slow) */
  typedef typename RsMaskT<K32>::type set_t;
  set_t free_rbg = R >= (int)(8 * sizeof(set_t)) ? ~(set_t)0 : (set_t)(((set_t)1 << R) - 1);
  int left = lane < S ? m->quota[lane] : 0; /* lane s: quota[s] - granted[s] */
  int n_taken = 0;
#ifdef RS_STAMPS
  int n_vec = 0;
#endif
  m->maskA[lane] = 0ull;
  m->maskB[lane] = 0ull;
  ((unsigned char*)m->n_level)[lane] = 0xFF;
  int n = N, pos = 0;
  while (n_taken < R && pos < n) {
    rs_scan_decide_vector<set_t>(s_sorted, pos, n, m, free_rbg, left, n_taken);
#ifdef RS_STAMPS
    ++n_vec;
#endif
    pos += 64;
    if (n_taken < R && n - pos > 64) {
      /* keep what is still live of the rest (in place: the writes trail the reads; eight chunks of reads in flight) */
      const set_t open_sl = (set_t)__ballot(left > 0);
      int kept = 0;
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
          const bool inb = x < n, lv = ((free_rbg >> rbg) & (open_sl >> sl) & 1) != 0;
          const unsigned long long mk = __builtin_amdgcn_ballot_w64(inb) & __builtin_amdgcn_ballot_w64(lv);
          const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
          if (inb & lv) s_sorted[kept + below] = e[j];
          kept += __popcll(mk);
        }
      }
      n = kept;
      pos = 0;
    }
  }
#ifdef RS_STAMPS
  if (threadIdx.x == 0) {
    stamp_acc[9] += (unsigned long long)n_taken;
    stamp_acc[10] += (unsigned long long)n_vec; /* vectors decided */
  }
#endif
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (lane < S) got = m->quota[lane] - left;
  const int o = ((unsigned char*)m->n_level)[lane];
  return (lane < R && o != 0xFF) ? o : -1; /* lane r: slice that got RBG r */
}

/*
 * VogelApproximate's two searches, without walking the row / the column.  One search is the reference's sequential scan
 *     for every allowed element in index order:  if (best < 0 || key > best) { best = key; arg = index; }
 *                                                else if (second < 0 || key > second) second = key;
 * (`second` is the largest key that was NOT a new best when it was met -- a new best does not demote the old one).  An element is
 * a new best exactly when no allowed element before it has a key at or above its own, i.e. when it is the first element of the set
 * {key >= its key}.  With the keys of the row (column) held as four bit planes over the element index -- keys are CQIs, 0..15 --
 * the elements of key v are one AND of four masks, and walking v = 15 .. 0 with `seen` = elements of a key >= v so far gives
 *     best   = the first v that has an element,   arg = its lowest index,
 *     second = the first v that has an element other than the lowest bit of `seen`   (that one is the new best among them),
 * usually within two or three values of v (a row holds 20 keys, a column up to 64, of ~9 distinct values): ~15 instructions per
 * value instead of ~6 per element, and no LDS traffic in the rounds.  tests/test_vogel_buckets.py checks the identity against the
 * sequential scan on random rows; the device is checked against the CPU restatement of this policy, which is pinned to the reference's own
 * unit code.
 */
template <class M>
struct RsKeyPlanes {
  M p[4]; /* bit b of element i's key at bit i of p[b] */
  M all;  /* the elements that exist */
};

template <class M>
__device__ __forceinline__ void vogel_best_second(const RsKeyPlanes<M>& pl, M allowed, bool part, int& best, int& second, int& arg) {
  best = second = arg = -1;
  const M cand = pl.all & allowed;
  M seen = 0;
  bool done = !part || cand == 0;
#pragma unroll
  for (int v = 15; v >= 0; --v) {
    if (__ballot(!done) == 0ull) break; /* wave-uniform */
    M A = cand;
#pragma unroll
    for (int b = 0; b < 4; ++b) A &= ((v >> b) & 1) ? pl.p[b] : (M)~pl.p[b];
    seen |= A;
    const M first = seen & (M)((M)0 - seen); /* the first element of {key >= v}: the only possible new best of this value */
    if (!done) {
      if (A != 0 && best < 0) {
        best = v;
        arg = (sizeof(M) == 8 ? __ffsll((long long)A) : __ffs((int)A)) - 1;
      }
      if ((A & ~first) != 0 && second < 0) second = v;
      done = (best >= 0 && second >= 0) || seen == cand;
    }
  }
}

template <int S_T, int R_T>
__device__ __forceinline__ int interslice_vogel(const uint32_t* s_elems, const RsMisc* m, int S_rt, int R_rt, int& got) {
  const int lane = lane_id();
  const int S = S_T ? S_T : S_rt, R = R_T ? R_T : R_rt; /* front-end constants in a shape-specialised build */
  /* VogelApproximate, ref: downlink-transport-scheduler.cpp:378-451.  R rounds; in each one every free RBG (lanes = RBGs)
   * looks for its best and "second" slice among the slices under quota, every such slice (lanes = slices) for its best and
   * "second" free RBG, and the candidate with the largest difference gets assigned.  Three details of the reference are
   * kept: (1) a new best does not demote the old best to second -- second is the largest value that was not a new best
   * when it was met; (2) `max_diff` is an int: a candidate wins when its (double) difference exceeds the TRUNCATED
   * running maximum, so the LAST candidate above the running truncated maximum wins, horizontal candidates (RBG
   * ascending) before vertical ones (slice ascending); (3) comparisons are on efficiencies, which are strictly
   * increasing in the CQI key (0 = no user), so keys are compared and only the differences use the doubles. */
  const int quota = lane < S ? m->quota[lane] : 0;
  int my_slice = -1; /* lane r: slice that got RBG r */
  /* the records do not change during the rounds: every RBG lane turns its row, every slice lane its column into bit planes once
   * per TTI (the columns through ballots over the RBG lanes: no second pass over the records) */
  typedef typename RsMaskT<(S_T != 0 && S_T <= 32)>::type rmask_t; /* over slices */
  typedef typename RsMaskT<(R_T != 0 && R_T <= 32)>::type cmask_t; /* over RBGs */
  RsKeyPlanes<rmask_t> row;
  RsKeyPlanes<cmask_t> col;
#pragma unroll
  for (int b = 0; b < 4; ++b) { row.p[b] = 0; col.p[b] = 0; }
  row.all = S >= (int)(8 * sizeof(rmask_t)) ? (rmask_t)~(rmask_t)0 : (rmask_t)(((rmask_t)1 << S) - 1);
  col.all = R >= (int)(8 * sizeof(cmask_t)) ? (cmask_t)~(cmask_t)0 : (cmask_t)(((cmask_t)1 << R) - 1);
  {
    const uint32_t* rowp = s_elems + (lane < R ? lane : 0) * S;
    for (int k0 = 0; k0 < S; k0 += 4) {
      uint32_t e4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) e4[q] = rowp[k0 + q < S ? k0 + q : S - 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + q;
        if (k < S) { /* wave-uniform */
          const int key = lane < R ? (int)(e4[q] >> 16) : 0;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const bool bit = ((key >> b) & 1) != 0;
            if (bit) row.p[b] |= (rmask_t)1 << k;
            const unsigned long long mk = __ballot(bit);
            if (lane == k) col.p[b] = (cmask_t)mk;
          }
        }
      }
    }
  }
  for (int round = 0; round < R; ++round) {
    const unsigned long long elig = __ballot(lane < S && got < quota);
    const unsigned long long freeb = __ballot(lane < R && my_slice < 0);
    const bool hpart = (freeb >> lane) & 1ull, vpart = (elig >> lane) & 1ull;
    /* horizontal search: lane j = free RBG j over the slices under quota; vertical: lane k = such a slice over the free RBGs */
    int h1, h2, hs, v1, v2, vr;
    vogel_best_second<rmask_t>(row, (rmask_t)elig, hpart, h1, h2, hs);
    vogel_best_second<cmask_t>(col, (cmask_t)freeb, vpart, v1, v2, vr);
    const double hd = (h1 < 0 ? -1.0 : m->eff16[h1]) - (h2 < 0 ? -1.0 : m->eff16[h2]);
    const double vd = (v1 < 0 ? -1.0 : m->eff16[v1]) - (v2 < 0 ? -1.0 : m->eff16[v2]);
    /* running truncated maximum before each candidate (-1 at the start), candidates in the reference's order */
    const int ht = hpart ? (int)hd : -1, vt = vpart ? (int)vd : -1;
    const int hinc = wave_scan_max_incl(ht);
    int hexc = __shfl_up(hinc, 1, 64);
    if (lane == 0) hexc = -1;
    const int hall = __builtin_amdgcn_readlane(hinc, 63);
    const int vinc = wave_scan_max_incl(vt);
    int vexc = __shfl_up(vinc, 1, 64);
    if (lane == 0) vexc = -1;
    if (vexc < hall) vexc = hall;
    if (hexc < -1) hexc = -1;
    if (vexc < -1) vexc = -1;
    const unsigned long long hacc = __ballot(hpart && hd > (double)hexc);
    const unsigned long long vacc = __ballot(vpart && vd > (double)vexc);
    int pick_rbg = -1, pick_slice = -1;
    if (vacc) {
      const int k = 63 - __clzll((long long)vacc);
      pick_slice = k;
      pick_rbg = __builtin_amdgcn_readlane(vr, k);
    } else if (hacc) {
      const int j = 63 - __clzll((long long)hacc);
      pick_rbg = j;
      pick_slice = __builtin_amdgcn_readlane(hs, j);
    }
    if (pick_rbg < 0 || pick_slice < 0) break; /* reference: uninitialised coordinates (undefined behaviour) */
    if (lane == pick_rbg) my_slice = pick_slice;
    if (lane == pick_slice) got++;
  }
  return my_slice;
}

/* um: RS_UMAP_SCRATCH_BYTES of LDS for rs_umap_order */
template <int S_T, int R_T>
__device__ __forceinline__ int interslice_subopt(const uint32_t* s_elems, const RsMisc* m, uint8_t* um, int S_rt, int R_rt, int& got) {
  const int lane = lane_id();
  const int S = S_T ? S_T : S_rt, R = R_T ? R_T : R_rt; /* front-end constants in a shape-specialised build */
  /* SubOpt, ref: downlink-transport-scheduler.cpp:274-349.  Every RBG (lanes = RBGs) starts at its best slice (first
   * maximum); then one RBG per round moves from a slice above its quota to a slice below it -- the move with the
   * smallest efficiency loss, first in (RBG ascending, `slice_fewer` iteration order) among equal losses.  That order
   * is libstdc++'s unordered_map order after the ascending insertions (erasures keep it): lane 0 computes it once per
   * TTI (rs_umap_order), lane p then holds the p-th key.  Counters live in lanes = slices; negative quotas count as 0. */
  int my_slice = -1;
  const uint32_t* row = s_elems + (lane < R ? lane : 0) * S;
  {
    int bestk = -1;
    for (int k = 0; k < S; ++k) {
      const int key = (int)(row[k] >> 16);
      if (key > bestk) { bestk = key; my_slice = k; }
    }
    if (lane >= R) my_slice = -1;
  }
  for (int sl = 0; sl < S; ++sl) {
    const int cnt = __popcll(__ballot(my_slice == sl));
    if (lane == sl) got = cnt;
  }
  int quota = lane < S ? m->quota[lane] : 0;
  quota = quota < 0 ? 0 : quota;
  int more = (lane < S && got > quota) ? got - quota : 0;
  int fewer = (lane < S && got < quota) ? quota - got : 0;
  unsigned long long more_mask = __ballot(more > 0), fewer_mask = __ballot(fewer > 0);
  int n_ord = 0;
  /* every lane runs it (wave-uniform input, the same bytes to the same LDS addresses): no divergent region around the serial walk */
  if (more_mask && fewer_mask) n_ord = rs_umap_order(fewer_mask, um, um + 68, um + 196);
  n_ord = __builtin_amdgcn_readfirstlane(n_ord);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int ordv = lane < n_ord ? um[196 + lane] : 0;
  /* A shape-specialised build keeps the efficiencies of its RBG's S records in registers and walks the slices in index order:
   * "first in the hashtable's order among equal losses" becomes "smallest position in that order" (rankv: lane k holds the
   * position of slice k), so the rounds read no LDS beyond the mover's new efficiency. */
  constexpr bool kRegs = S_T != 0 && S_T <= 32;
  double effk[kRegs ? S_T : 1];
  int rankv = 255;
  double own_eff = 0.0;
  if constexpr (kRegs) {
#pragma unroll
    for (int k = 0; k < S_T; ++k) effk[k] = m->eff16[row[k] >> 16];
    um[68 + lane] = 255; /* the bucket scratch of rs_umap_order is free again */
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < n_ord) um[68 + ordv] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    rankv = um[68 + lane];
    own_eff = m->eff16[row[my_slice >= 0 ? my_slice : 0] >> 16];
  }
  while (more_mask && fewer_mask) {
    const bool cand = my_slice >= 0 && ((more_mask >> my_slice) & 1ull);
    double least = 1.7976931348623157e308;
    int to_sl = -1;
    if constexpr (kRegs) {
      int best_rank = 1 << 30;
#pragma unroll
      for (int k = 0; k < S_T; ++k)
        if ((fewer_mask >> k) & 1ull) {
          const int rk = __builtin_amdgcn_readlane(rankv, k);
          const double loss = own_eff - effk[k];
          if (loss < least || (loss == least && rk < best_rank)) { least = loss; to_sl = k; best_rank = rk; }
        }
    } else {
      own_eff = m->eff16[row[my_slice >= 0 ? my_slice : 0] >> 16];
      for (int q = 0; q < n_ord; ++q) {
        const int key = __builtin_amdgcn_readlane(ordv, q);
        if (!((fewer_mask >> key) & 1ull)) continue;
        const double loss = own_eff - m->eff16[row[key] >> 16];
        if (loss < least) { least = loss; to_sl = key; }
      }
    }
    /* smallest loss over the candidate RBGs, lowest RBG among equals: losses are >= 0, their bit patterns order
     * like the values */
    const long long lb = __double_as_longlong(least);
    const int hi = cand ? (int)(lb >> 32) : 0x7fffffff;
    const int hmin = wave_min(hi);
    const int lo = (cand && hi == hmin) ? (int)((uint32_t)lb ^ 0x80000000u) : 0x7fffffff;
    const int lmin = wave_min(lo);
    const unsigned long long hit = __ballot(cand && hi == hmin && lo == lmin);
    if (!hit) break; /* reference asserts */
    const int rbg = __ffsll((long long)hit) - 1;
    const int from = __builtin_amdgcn_readlane(my_slice, rbg), to = __builtin_amdgcn_readlane(to_sl, rbg);
    if (lane == rbg) my_slice = to;
    if constexpr (kRegs) {
      const double to_eff = m->eff16[row[to] >> 16]; /* every lane reads (uniform `to`), the mover keeps it */
      if (lane == rbg) own_eff = to_eff;
    }
    if (lane == from) { got--; more--; }
    if (lane == to) { got++; fewer--; }
    more_mask = __ballot(more > 0 && got > 0);
    fewer_mask = __ballot(fewer > 0);
  }
  return my_slice;
}

}  // namespace

#endif /* RS_INTERSLICE_H_ */
