/*
 * rs_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the RadioSaber downlink RBG allocation path.
 *
 * One workgroup = one cell.  The workgroup keeps the cell's whole scheduling state in LDS (per-RBG
 * CQI grid u8[U][R], PF averages f64[U], counters, the CQI->rate / EESM tables, slice quotas) and
 * runs n_ttis complete DoSchedule() iterations back to back:
 *
 *   P0  CQI refresh (every 40 TTIs)        HBM -> LDS, 16 B per lane, coalesced
 *   P1  PF EWMA update per user            ref: src/flows/radio-bearer.cpp:139-164
 *   P2  slice quotas (lanes = slices)      ref: downlink-transport-scheduler.cpp:463-521
 *   P3  best user per (RBG, slice)         ref: :530-567   (the UE x RBG metric scan, FP64 division)
 *   P4  inter-slice assignment             ref: :249-272 GreedyByRow / :351-376 MaximizeCell
 *   P5  apply + EESM link adaptation + DoStopSchedule counters     ref: :589-674, :170-221
 *
 * No MFMA: the only matrix-shaped object (metric[R][U]) is consumed by an argmax.  All floating
 * point is IEEE FP64 add/mul/div in the reference's operation order; the file MUST be compiled with
 * -ffp-contract=off (the reference is x86-64 SSE2 without FMA).  libm never runs on the device: the
 * host evaluates the transcendental tables (rs_link_tables) and the device only compares.
 *
 * `ref:` paths are relative to /root/reference/src/protocolStack/mac/packet-scheduler/ unless they
 * start with src/.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rs_amc_tables.inc"
#include "rs_device.h"
#include "rs_sort_emul.h"

/* 3GPP TS 36.213 Table 7.1.7.2.1-1 (110 x 27 ints, 11.9 KB): read through the vector cache, a few
 * lookups per TTI */
__device__ const int32_t d_tbs_table[110 * 27] = {RS_AMC_TBS_TABLE};

namespace {

typedef RsMisc Misc;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    int w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}

/* glibc TYPE_3 rand(): ring of 31 words held one per lane of wave 0 (lane l = r[l]); f, b uniform.
 * (glibc 2.35 stdlib/random_r.c __random_r; the reference draws from libc rand():
 *  downlink-transport-scheduler.cpp:490,511) */
struct WaveRng {
  uint32_t r; /* this lane's ring word */
  int f, b;   /* wave-uniform */
  __device__ __forceinline__ int next() {
    uint32_t vf = __builtin_amdgcn_readlane(r, f);
    uint32_t vb = __builtin_amdgcn_readlane(r, b);
    uint32_t v = vf + vb;
    r = ((int)(threadIdx.x & 63) == f) ? v : r;
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    return (int)(v >> 1);
  }
};

/* ref: src/protocolStack/mac/AMCModule.cpp:306-317 incl. the as-shipped -O0 out-of-bounds rule
 * T[-1][i], carried in tab->tbs_row_m1 */
__device__ __forceinline__ int tbs_bits(int itbs, int nprb, const int32_t* row_m1) {
  if (nprb <= 110) return d_tbs_table[(nprb - 1) * 27 + itbs];
  int sub = nprb / 5, rest = nprb % 5;
  int tail = rest == 0 ? row_m1[itbs] : d_tbs_table[(rest - 1) * 27 + itbs];
  return 5 * d_tbs_table[(sub - 1) * 27 + itbs] + tail;
}

struct LdsArr {
  uint32_t* p;
  __device__ __forceinline__ uint32_t& operator[](int i) { return p[i]; }
};
struct LdsInt {
  int32_t* p;
  __device__ __forceinline__ int32_t& operator[](int i) { return p[i]; }
};

}  // namespace

/*
 * std::__introsort_loop, wave-parallel.  Every recursion level's sub-ranges are disjoint, so they
 * are queued per level and each wave partitions whole sub-ranges; a partition is two passes over
 * the sub-range in 64-element chunks:
 *   pass 1  ballots of "left-scan stop" (key <= pivot) and "right-scan stop" (key >= pivot) on the
 *           array before the partition; prefix popcounts turn them into the stop lists L (ascending)
 *           and Rr (via posB, read backwards), kept in LDS scratch;
 *   pass 2  the serial Hoare loop performs exactly the swaps (L[j], Rr[j]) for j < k, k = first j
 *           with L[j] >= Rr[j], and returns L[0] if k == 0 else min(L[k], Rr[k-1])
 *           (derivation: DESIGN.md; CPU model: tests/test_partition_model.py).
 * The median-of-3 pivot move and the (never observed) heap-sort fallback at depth 0 run on one lane
 * with the serial code of rs_sort_emul.h.  All waves of the workgroup must call this.
 */
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ void introsort_loop_parallel(uint32_t* v, int N, uint16_t* posA, uint16_t* posB, Misc* m) {
  const int lane = lane_id(), wave = wave_id(), nwaves = blockDim.x >> 6;
  if (threadIdx.x < 48) m->n_level[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0 && N > 16) {
    m->n_level[0] = 1;
    m->q_first[0][0] = 0;
    m->q_last[0][0] = (int16_t)N;
    m->q_depth[0][0] = (int16_t)(2 * rs_sort::floor_log2(N));
  }
  __syncthreads();
  for (int level = 0; level < 47; ++level) {
    const int par = level & 1;
    const int n_cur = m->n_level[level];
    if (n_cur == 0) break;
    for (int i = wave; i < n_cur; i += nwaves) {
      const int first = m->q_first[par][i], last = m->q_last[par][i];
      int depth = m->q_depth[par][i];
      if (depth == 0) {
        if (lane == 0) {
          LdsArr a{v};
          rs_sort::heap_sort(a, first, last);
        }
        continue;
      }
      --depth;
      if (lane == 0) {
        LdsArr a{v};
        rs_sort::median_to_first(a, first, first + 1, first + (last - first) / 2, last - 1);
      }
      wave_lds_sync();
      const int pk = (int)(v[first] >> 16);
      int cntA = 0, cntB = 0;
      const unsigned long long lt = (1ull << lane) - 1ull;
      for (int base = first + 1; base < last; base += 64) {
        const int x = base + lane;
        const bool in = x < last;
        const int k = in ? (int)(v[x] >> 16) : 0;
        const bool isA = in && k <= pk;
        const bool isB = in && k >= pk;
        const unsigned long long mA = __ballot(isA), mB = __ballot(isB);
        if (isA) posA[first + cntA + __popcll(mA & lt)] = (uint16_t)x;
        if (isB) posB[first + cntB + __popcll(mB & lt)] = (uint16_t)x;
        cntA += __popcll(mA);
        cntB += __popcll(mB);
      }
      wave_lds_sync();
      const int nmin = cntA < cntB ? cntA : cntB;
      int k = 0;
      for (int j0 = 0; j0 < nmin; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < nmin;
        const int l = valid ? (int)posA[first + j] : 0;
        const int r = valid ? (int)posB[first + cntB - 1 - j] : 0;
        const bool sw = valid && l < r;
        const unsigned long long ms = __ballot(sw);
        if (sw) {
          const uint32_t a = v[l], b = v[r];
          v[l] = b;
          v[r] = a;
        }
        k += __popcll(ms);
        if (ms != __ballot(valid)) break;
      }
      wave_lds_sync();
      int cut;
      if (k == 0) {
        cut = posA[first];
      } else {
        const int lk = k < cntA ? (int)posA[first + k] : (1 << 30);
        const int rk = (int)posB[first + cntB - k]; /* Rr[k-1] */
        cut = lk < rk ? lk : rk;
      }
      if (lane == 0) {
        if (cut - first > 16) {
          int s = atomicAdd(&m->n_level[level + 1], 1);
          m->q_first[par ^ 1][s] = (int16_t)first;
          m->q_last[par ^ 1][s] = (int16_t)cut;
          m->q_depth[par ^ 1][s] = (int16_t)depth;
        }
        if (last - cut > 16) {
          int s = atomicAdd(&m->n_level[level + 1], 1);
          m->q_first[par ^ 1][s] = (int16_t)cut;
          m->q_last[par ^ 1][s] = (int16_t)last;
          m->q_depth[par ^ 1][s] = (int16_t)depth;
        }
      }
    }
    __syncthreads();
  }
}

#ifdef RS_STAMPS
/* diagnostic build only: cycles per phase of thread 0, accumulated over the launch (never in the
 * product library; the values go to a buffer nothing else reads) */
#define RS_STAMP(i)                                                \
  do {                                                             \
    if (tid == 0) {                                                \
      unsigned long long now_ = __builtin_readcyclecounter();      \
      stamp_acc[i] += now_ - stamp_prev;                           \
      stamp_prev = now_;                                           \
    }                                                              \
  } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif

template <int SCHED>
__global__ void __launch_bounds__(1024) rs_cell_kernel(RsLaunch p) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int cell = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lane = lane_id(), wave = wave_id();
  const int S = p.S, U = p.U, R = p.R, G = p.G;
  constexpr bool kTransport = (SCHED == 8 || SCHED == 9);

  double* s_avg = (double*)lds;
  double* s_avgk = (double*)(lds + p.off_avgk);
  int32_t* s_tx = (int32_t*)(lds + p.off_tx);
  int64_t* s_cumb = (int64_t*)(lds + p.off_cumb);
  int32_t* s_cumr = (int32_t*)(lds + p.off_cumr);
  double* s_num = (double*)(lds + p.off_tab); /* metric numerator per CQI */
  double* s_e = s_num + 16;
  double* s_x = s_e + 16;
  double* s_w = (double*)(lds + p.off_slice);
  double* s_sstate = s_w + 64;
  uint16_t* s_best_user = (uint16_t*)(lds + p.off_items);
  double* s_best_metric = (double*)(lds + p.off_elems); /* sched 1 only (aliases elems) */
  uint32_t* s_elems = (uint32_t*)(lds + p.off_elems);
  uint32_t* s_sorted = (uint32_t*)(lds + p.off_sorted);
  Misc* m = (Misc*)(lds + p.off_misc);
  uint8_t* s_cqi = lds + p.off_cqi;

  const RsTables* tab = p.tab;
  RsCellScalars* scal = p.scal + cell;

  /* ---------------- load the cell ---------------- */
  for (int u = tid; u < U; u += nt) {
    s_avg[u] = p.avg[(size_t)cell * U + u];
    s_tx[u] = p.tx_bytes[(size_t)cell * U + u];
    s_cumb[u] = 0;
    s_cumr[u] = 0;
  }
  if (tid < 16) {
    s_num[tid] = SCHED == 1 ? tab->pfnum[tid] : tab->kbps[tid];
    s_e[tid] = tab->eesm_e[tid];
    s_x[tid] = tab->eesm_x[tid];
  }
  if (tid < S) {
    s_w[tid] = p.weight[tid];
    s_sstate[tid] = p.slice_state[(size_t)cell * S + tid];
  }
  /* segments scanned in P3: slices (7/8/9) or fixed runs of RS_PF_SEG users (1) */
  if (SCHED == 1) {
    if (tid <= p.n_seg) m->seg_begin[tid] = min(tid * RS_PF_SEG, U);
  } else {
    if (tid <= S) m->seg_begin[tid] = U; /* filled below */
  }
  __syncthreads();
  if (SCHED != 1) {
    /* user_slice is non-decreasing: slice s = [first u with slice >= s, ...) */
    for (int u = tid; u < U; u += nt) {
      int s = p.user_slice[u];
      int sp = u == 0 ? -1 : (int)p.user_slice[u - 1];
      for (int q = sp + 1; q <= s; q++) m->seg_begin[q] = u;
    }
  }
  double t = scal->t;
  double last_update = scal->last_update;
  long long last_sent = scal->last_sent;
  int reported = scal->reported;
  int served_prev = scal->served_prev;
  long long n_done = scal->n_done;
  WaveRng rng;
  rng.r = 0;
  rng.f = scal->rng_f;
  rng.b = scal->rng_b;
  if (wave == 0 && lane < 31) rng.r = scal->rng_r[lane];
  const int nb_rbs = R * G;
  int local_err = 0;
  __syncthreads();

#ifdef RS_STAMPS
  unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_readcyclecounter();
#endif
  for (int tti = 0; tti < p.n_ttis; ++tti) {
    RS_STAMP(11);
    /* ---------------- P0: CQI refresh ---------------- */
    if (p.cqi_mode == RS_CQI_EPOCHS) {
      if (p.direct || n_done % p.refresh == 0) {
        long long e = p.direct ? 0 : n_done / p.refresh;
        if (e >= p.n_epochs) { local_err = RS_CQI_EPOCHS; e = p.n_epochs - 1; }
        const uint4* src = (const uint4*)(p.epochs + ((size_t)cell * p.n_epochs + (size_t)e) * p.grid_stride);
        uint4* dst = (uint4*)s_cqi;
        const int n16 = (int)(p.grid_stride >> 4);
        for (int i = tid; i < n16; i += nt) dst[i] = src[i];
      }
    } else if (p.cqi_mode == RS_CQI_TRACE) {
      /* ref: src/device/CqiManager/cqi-manager.cpp:105-123 (interval 40),
       *      src/protocolStack/mac/enb-mac-entity.cc:189-191 */
      if (!reported || ((int)(t * 1000) - last_sent) >= 40) {
        reported = 1;
        last_sent = (long long)(t * 1000);
        int stamp = (int)(t * 1000 / 40);
        int row = stamp % p.row_mod;
        if (row >= p.n_rows) { local_err = RS_CQI_TRACE; row = 0; }
        for (int i = tid; i < U * R; i += nt) {
          int u = i / R, r = i - u * R;
          int tr = p.user_trace[(size_t)cell * U + u];
          s_cqi[i] = p.trace[((size_t)tr * p.n_rows + row) * R + r];
        }
      }
    }
    /* ---------------- P1: PF EWMA (ref: src/flows/radio-bearer.cpp:139-164) ---------------- */
    if (!p.direct && !(t == last_update)) {
      const double dt = t - last_update;
      for (int u = tid; u < U; u += nt) {
        double rate = (double)(s_tx[u] * 8) / dt;
        const double beta = 0.02;
        double a = ((1 - beta) * s_avg[u]) + (beta * rate);
        if (a < 1) a = 1;
        s_avg[u] = a;
        s_tx[u] = 0;
      }
    }
    if (!p.direct) last_update = t;
    if (SCHED != 1) {
      /* ref: :685-689  averageRate = 1 + sum(avg); averageRate /= 1000.0 */
      for (int u = tid; u < U; u += nt) {
        double a = 1;
        a += s_avg[u];
        a /= 1000.0;
        s_avgk[u] = a;
      }
    }
    RS_STAMP(0);
    int seg_lo = 0; /* NVS: the served slice */
    /* ---------------- P2: quotas / slice choice (wave 0, lanes = slices) ---------------- */
    if (wave == 0) {
      int r0 = p.rand0, r1 = p.rand1;
      if (!p.direct && kTransport) {
        if (p.phy_draws)
          for (int i = 0; i < served_prev; i++) (void)rng.next();
        r0 = rng.next();
        r1 = rng.next();
      } else if (!p.direct && p.phy_draws) {
        for (int i = 0; i < served_prev; i++) (void)rng.next();
      }
      if (kTransport) {
        const bool in = lane < S;
        const bool has = in && (m->seg_begin[lane + 1] > m->seg_begin[lane]);
        const int nonempty = __popcll(__ballot(has));
        int target = 0;
        if (has) target = (int)(nb_rbs * s_w[lane] + s_sstate[lane]);
        int extra = nb_rbs - wave_sum(target);
        /* first non-empty slice in the rotation starting at rand % S */
        int pos0 = has ? (int)(((long long)lane - (r0 % S) + S) % S) : 1 << 20;
        int first0 = wave_min(pos0);
        if (has) {
          target += extra / nonempty;
          if (pos0 == first0) target += extra % nonempty;
        }
        int quota = in ? (int)(target / G) : 0;
        int extra_g = R - wave_sum(quota);
        int pos1 = has ? (int)(((long long)lane - (r1 % S) + S) % S) : 1 << 20;
        int first1 = wave_min(pos1);
        if (has) {
          quota += extra_g / nonempty;
          if (pos1 == first1) quota += extra_g % nonempty;
        }
        if (lane < 64) {
          m->target[lane] = target;
          m->quota[lane] = quota;
          m->got[lane] = 0;
          m->final_rbgs[lane] = 0;
        }
      } else if (SCHED == 7) {
        /* SelectSliceToServe, ref: downlink-nvs-scheduler.cpp:94-142 */
        int pick;
        if (p.direct) {
          pick = 0; /* the caller passes only the served slice's users */
        } else {
          const bool in = lane < S;
          const bool has = in && (m->seg_begin[lane + 1] > m->seg_begin[lane]);
          double ew = in ? s_sstate[lane] : 1.0;
          unsigned long long zero = __ballot(has && ew == 0);
          unsigned long long hasm = __ballot(has);
          int first_zero = zero ? __ffsll((long long)zero) - 1 : 64;
          /* scan order: slices before the first zero-ewma slice compete with '>=' (last max wins),
           * but a zero-ewma slice ends the scan and wins outright */
          if (zero) {
            pick = first_zero;
          } else {
            double score = has ? s_w[lane] / ew : -1.0;
            /* argmax, ties -> highest lane */
            double best = score;
            int bl = has ? lane : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              double ob = __shfl_xor(best, o, 64);
              int ol = __shfl_xor(bl, o, 64);
              if (ob > best || (ob == best && ol > bl)) { best = ob; bl = ol; }
            }
            pick = hasm ? (bl < 0 ? 0 : bl) : 0;
          }
          const double beta = 0.01;
          if (has) {
            double e2 = (1 - beta) * ew;
            if (lane == pick) e2 += beta * 1;
            s_sstate[lane] = e2;
          }
        }
        if (lane == 0) m->nvs_slice = pick;
        if (lane < 64) { m->target[lane] = 0; m->quota[lane] = 0; }
      } else {
        if (lane < 64) { m->target[lane] = 0; m->quota[lane] = 0; }
      }
    }
    __syncthreads();
    RS_STAMP(1);
    if (SCHED == 7) seg_lo = p.direct ? 0 : m->nvs_slice;

    /* ---------------- P3: best user of every (RBG, segment) ---------------- */
    {
      const int n_items = p.n_items;
      const int n_seg = p.n_seg;
      for (int it = tid; it < n_items; it += nt) {
        int sg = it / R, r = it - sg * R; /* r fastest: neighbouring lanes read neighbouring CQI bytes */
        int seg = SCHED == 7 ? seg_lo : sg;
        int ub = m->seg_begin[seg], ue = m->seg_begin[seg + 1];
        if (SCHED == 7 && p.direct) { ub = 0; ue = U; }
        double best = SCHED == 1 ? 0.0 : (SCHED == 7 ? -1.7976931348623157e308 : -1.0);
        int bu = -1, bkey = 0;
        int sl_eps = 1, sl_psi = 1;
        if (SCHED != 1) {
          int sl = SCHED == 7 ? (p.direct ? (int)p.user_slice[0] : seg) : seg;
          sl_eps = p.eps[sl];
          sl_psi = p.psi[sl];
        }
        for (int u = ub; u < ue; ++u) {
          int c = s_cqi[u * R + r];
          double metric;
          if (SCHED == 1) {
            /* ref: dl-pf-packet-scheduler.cpp:128-140  (se*180000.)/avg */
            metric = s_num[c] / s_avg[u];
          } else {
            /* ref: :688-693  pow(se_kbps, eps) / pow(avg_kbps, psi), eps, psi in {0,1} */
            double num = sl_eps ? s_num[c] : 1.0;
            double den = sl_psi ? s_avgk[u] : 1.0;
            metric = num / den;
          }
          if (metric > best) { best = metric; bu = u; bkey = c; }
        }
        s_best_user[it] = (uint16_t)bu;
        if (kTransport) {
          /* MaximizeCell's vector is RBG-major, slice-minor (:357-360) */
          s_elems[r * S + sg] = ((uint32_t)bkey << 16) | ((uint32_t)r << 8) | (uint32_t)sg;
        } else if (SCHED == 1) {
          s_best_metric[it] = best;
        }
      }
      (void)n_seg;
    }
    __syncthreads();
    RS_STAMP(2);

    /* ---------------- P4: inter-slice assignment ---------------- */
    if (SCHED == 8) {
      /* GreedyByRow, ref: :249-272 -- RBG ascending, argmax eff over slices under quota, first max
       * wins.  eff is strictly increasing in CQI (0 for an empty slice), so integer keys compare alike. */
      if (wave == 0) {
        int got = 0;
        const int quota = lane < S ? m->quota[lane] : 0;
        for (int r = 0; r < R; ++r) {
          int key = lane < S ? (int)(s_elems[r * S + lane] >> 16) : -1;
          bool ok = lane < S && got < quota;
          int packed = ok ? (key << 6) | (63 - lane) : -1;
          int bestp = wave_max(packed);
          int sl = bestp < 0 ? -1 : 63 - (bestp & 63);
          if (lane == sl) got++;
          if (lane == 0) m->rbg_slice[r] = sl;
        }
      }
    } else if (SCHED == 9) {
      const int N = R * S;
      /* std::sort emulation, step 1: the introsort loop */
#ifdef RS_SERIAL_SORT
      if (tid == 0) {
        LdsArr a{s_elems};
        LdsInt st{m->stack};
        rs_sort::introsort_loop(a, N, st);
      }
      __syncthreads();
#else
      introsort_loop_parallel(s_elems, N, (uint16_t*)s_sorted, (uint16_t*)s_sorted + N, m);
#endif
      RS_STAMP(3);
      /* step 2: final insertion sort == stable counting sort by descending key (wave 0) */
      if (wave == 0) {
        int base = 0; /* lane q (< 16): output offset of key q */
        {
          int cnt = 0;
          for (int c0 = 0; c0 < N; c0 += 64) {
            int i = c0 + lane;
            int k = i < N ? (int)(s_elems[i] >> 16) : -1;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              unsigned long long mk = __ballot(k == q);
              if (lane == q) cnt += __popcll(mk);
            }
          }
          /* descending: offset(q) = sum of counts of keys > q */
          int tot = 0;
#pragma unroll
          for (int q = 15; q >= 0; --q) {
            int cq = __shfl(cnt, q, 64);
            if (lane == q) base = tot;
            tot += cq;
          }
        }
        for (int c0 = 0; c0 < N; c0 += 64) {
          int i = c0 + lane;
          uint32_t e = i < N ? s_elems[i] : 0;
          int k = i < N ? (int)(e >> 16) : -1;
          int my_base = __shfl(base, k < 0 ? 0 : k, 64);
          int rank = 0;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            unsigned long long mk = __ballot(k == q);
            if (k == q) rank = __popcll(mk & ((1ull << lane) - 1ull));
            if (lane == q) base += __popcll(mk);
          }
          if (i < N) s_sorted[my_base + rank] = e;
        }
        RS_STAMP(4);
        /* MaximizeCell greedy scan, ref: :362-369 */
        unsigned long long taken = 0;
        int assigned = 0;
        if (lane < R) m->rbg_slice[lane] = -1;
        for (int c0 = 0; c0 < N && assigned < R; c0 += 64) {
          int i = c0 + lane;
          uint32_t e = i < N ? s_sorted[i] : 0;
          int rbg = (e >> 8) & 63, sl = e & 63;
          while (true) {
            bool ok = i < N && !((taken >> rbg) & 1ull) && m->got[sl] < m->quota[sl];
            unsigned long long mk = __ballot(ok);
            if (!mk) break;
            int f = __ffsll((long long)mk) - 1;
            int frbg = __shfl(rbg, f, 64);
            if (lane == f) {
              m->rbg_slice[rbg] = sl;
              m->got[sl] += 1;
            }
            taken |= 1ull << frbg;
            assigned++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          }
        }
      }
    } else {
      /* sched 1 / 7: per RBG, first maximum over the segments in ascending order */
      if (tid < R) {
        const int r = tid;
        if (SCHED == 1) {
          double best = 0.0;
          int bu = -1;
          for (int sg = 0; sg < p.n_seg; ++sg) {
            double v = s_best_metric[sg * R + r];
            int u = s_best_user[sg * R + r];
            if (u != 0xFFFF && v > best) { best = v; bu = u; }
          }
          m->owner[r] = bu;
        } else {
          int u = s_best_user[r];
          m->owner[r] = u == 0xFFFF ? -1 : u;
        }
      }
    }
    RS_STAMP(5);
    __syncthreads();
    RS_STAMP(6);

    /* ---------------- P5: apply, link adaptation, accounting (wave 0, lanes = RBGs) ---------------- */
    if (wave == 0) {
      int owner = -1;
      if (lane < R) {
        if (kTransport) {
          int sl = m->rbg_slice[lane];
          if (sl >= 0) {
            int u = s_best_user[sl * R + lane];
            owner = u == 0xFFFF ? -1 : u;
            if (owner >= 0) atomicAdd(&m->final_rbgs[sl], 1);
          }
        } else {
          owner = m->owner[lane];
        }
        m->owner[lane] = owner;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      /* leader = lowest RBG of each served user */
      bool leader = owner >= 0;
      for (int r2 = 0; r2 < R; ++r2) {
        int o2 = __shfl(owner, r2, 64);
        if (r2 < lane && o2 == owner) leader = false;
      }
      unsigned long long lead_mask = __ballot(leader);
      int tbs = 0, nprb = 0, fcqi = 0, mcs = 0;
      if (leader) {
        /* ref: :638-651 -- PRBs in RBG-ascending order, G identical adds per RBG
         * (src/utility/eesm-effective-sinr.h:33-46 with the exp() values tabulated by the host) */
        double sum = 0;
        for (int r2 = lane; r2 < R; ++r2) {
          if (m->owner[r2] == owner) {
            double ev = s_e[s_cqi[owner * R + r2]];
            for (int k = 0; k < G; ++k) sum += ev;
            nprb += G;
          }
        }
        double x = sum / (double)nprb;
        if (x == 0) {
          fcqi = 15;
        } else {
          fcqi = 1;
#pragma unroll
          for (int k = 1; k <= 13; ++k) fcqi += (x <= s_x[k]) ? 1 : 0;
        }
        mcs = tab->mcs_of_cqi[fcqi];
        tbs = tbs_bits(tab->itbs_of_cqi[fcqi], nprb, tab->tbs_row_m1);
        /* DoStopSchedule, ref: :170-221 (bytes = bits/8, capped by dataToTransmit = 1e8) */
        int bytes = tbs / 8;
        if (bytes > 100000000) bytes = 100000000;
        if (bytes > 0) {
          s_tx[owner] += bytes;
          s_cumb[owner] += bytes;
          s_cumr[owner] += nprb;
        }
      }
      served_prev = __popcll(lead_mask);
      /* ref: :618-620 slice_rbs_offset_ = target - final_rbgs*rbg_size */
      if (kTransport && lane < S) s_sstate[lane] = (double)(m->target[lane] - m->final_rbgs[lane] * G);
      if (lane == 0) m->served = served_prev;
      /* optional log */
      if (p.log_map) {
        size_t row = (size_t)cell * p.n_ttis + tti;
        if (lane < R) p.log_map[row * R + lane] = (int16_t)owner;
        if (lane < S) {
          if (p.log_quota) p.log_quota[row * S + lane] = (int16_t)m->quota[lane];
          if (p.log_target) p.log_target[row * S + lane] = (int16_t)m->target[lane];
        }
        if (leader) {
          if (p.log_tbs) p.log_tbs[row * U + owner] = tbs;
          if (p.log_uinfo) p.log_uinfo[row * U + owner] = nprb | (fcqi << 16) | (mcs << 24);
        }
      }
    }
    RS_STAMP(7);
    __syncthreads();
    RS_STAMP(8);
    served_prev = m->served;
    n_done += 1;
    if (!p.direct) t += 0.001; /* ref: src/core/eventScheduler/simulator.cc:117-126 */
  }

  /* ---------------- store the cell ---------------- */
  for (int u = tid; u < U; u += nt) {
    p.avg[(size_t)cell * U + u] = s_avg[u];
    p.tx_bytes[(size_t)cell * U + u] = s_tx[u];
    p.cum_bytes[(size_t)cell * U + u] += s_cumb[u];
    p.cum_rbs[(size_t)cell * U + u] += s_cumr[u];
  }
  if (tid < S) p.slice_state[(size_t)cell * S + tid] = s_sstate[tid];
  if (wave == 0 && lane < 31) scal->rng_r[lane] = rng.r;
  if (tid == 0) {
    scal->t = t;
    scal->last_update = last_update;
    scal->last_sent = last_sent;
    scal->reported = reported;
    scal->served_prev = served_prev;
    scal->n_done = n_done;
    scal->rng_f = rng.f;
    scal->rng_b = rng.b;
    if (local_err) atomicExch(p.err, local_err);
#ifdef RS_STAMPS
    if (p.stamps)
      for (int i = 0; i < 12; ++i) p.stamps[(size_t)cell * 12 + i] = stamp_acc[i];
#endif
  }
}

/* ------------------------------------------------------------------------------------------
 * Synthetic CQI grids: i.i.d. draws from a 15-bin histogram, counter-based (SplitMix64 of
 * (seed, cell, epoch, user, rbg)).  One thread per byte, 16 consecutive bytes per lane.
 * ------------------------------------------------------------------------------------------ */
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

struct RsCdf {
  uint32_t c[16]; /* c[q] = floor(2^32 * P(cqi <= q+1)), q = 0..14 (c[14] = 2^32-1) */
};

__global__ void rs_synth_cqi_kernel(uint8_t* epochs, int64_t grid_stride, int n_cells, int n_epochs, int U,
                                    int R, uint64_t seed, RsCdf cdf) {
  const int64_t grids = (int64_t)n_cells * n_epochs;
  const int64_t per_grid16 = grid_stride >> 4;
  const int64_t total = grids * per_grid16;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
    int64_t g = w / per_grid16;
    int64_t o = (w - g * per_grid16) << 4;
    uint8_t out[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      int64_t idx = o + k;
      uint8_t v = 0;
      if (idx < (int64_t)U * R) {
        uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)g * 0x100000001B3ull + (uint64_t)idx));
        uint32_t x = (uint32_t)(h >> 32);
        int q = 0;
#pragma unroll
        for (int j = 0; j < 14; ++j) q += x > cdf.c[j] ? 1 : 0;
        v = (uint8_t)(q + 1);
      }
      out[k] = v;
    }
    *(uint4*)(epochs + g * grid_stride + o) = *(const uint4*)out;
  }
}

/* per-slice cumulative bytes over all cells -> d_out[S] (uint64) */
__global__ void rs_slice_bytes_kernel(const int64_t* cum_bytes, const uint8_t* user_slice, int n_cells, int U,
                                      int S, unsigned long long* d_out) {
  __shared__ unsigned long long acc[64];
  if (threadIdx.x < 64) acc[threadIdx.x] = 0;
  __syncthreads();
  const int64_t total = (int64_t)n_cells * U;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int u = (int)(i % U);
    atomicAdd(&acc[user_slice[u]], (unsigned long long)cum_bytes[i]);
  }
  __syncthreads();
  if (threadIdx.x < S && acc[threadIdx.x]) atomicAdd(&d_out[threadIdx.x], acc[threadIdx.x]);
}

/* host-callable launchers (defined here so that the kernels stay in one translation unit) */
extern "C" hipError_t rs_launch_cells(const RsLaunch* p, int threads, hipStream_t stream) {
  dim3 grid(p->n_cells), block(threads);
  switch (p->sched) {
    case 1: hipLaunchKernelGGL(rs_cell_kernel<1>, grid, block, p->lds_bytes, stream, *p); break;
    case 7: hipLaunchKernelGGL(rs_cell_kernel<7>, grid, block, p->lds_bytes, stream, *p); break;
    case 8: hipLaunchKernelGGL(rs_cell_kernel<8>, grid, block, p->lds_bytes, stream, *p); break;
    case 9: hipLaunchKernelGGL(rs_cell_kernel<9>, grid, block, p->lds_bytes, stream, *p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

extern "C" hipError_t rs_prepare_kernels(int max_lds_bytes) {
  hipError_t e;
  e = hipFuncSetAttribute((const void*)rs_cell_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)rs_cell_kernel<7>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)rs_cell_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void*)rs_cell_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes);
  return e;
}

extern "C" hipError_t rs_launch_synth(uint8_t* epochs, int64_t grid_stride, int n_cells, int n_epochs, int U, int R,
                                      uint64_t seed, const uint32_t* cdf16, hipStream_t stream) {
  RsCdf cdf;
  for (int i = 0; i < 16; ++i) cdf.c[i] = cdf16[i];
  int64_t total = (int64_t)n_cells * n_epochs * (grid_stride >> 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rs_synth_cqi_kernel, dim3(blocks), dim3(256), 0, stream, epochs, grid_stride, n_cells, n_epochs,
                     U, R, seed, cdf);
  return hipGetLastError();
}

extern "C" hipError_t rs_launch_slice_bytes(const int64_t* cum_bytes, const uint8_t* user_slice, int n_cells, int U,
                                            int S, unsigned long long* d_out, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * S, stream);
  if (e != hipSuccess) return e;
  int64_t total = (int64_t)n_cells * U;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rs_slice_bytes_kernel, dim3(blocks), dim3(256), 0, stream, cum_bytes, user_slice, n_cells, U, S,
                     d_out);
  return hipGetLastError();
}
