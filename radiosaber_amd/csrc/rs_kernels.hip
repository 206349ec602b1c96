/*
 * rs_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the RadioSaber downlink RBG allocation path.
 *
 * One workgroup = one cell.  The workgroup keeps the cell's whole scheduling state in LDS (RBG-major
 * CQI grid u8[R][Upad], PF averages f64[U], the CQI->rate / EESM / TBS tables, slice quotas) and runs
 * n_ttis complete DoSchedule() iterations back to back:
 *
 *   P0  CQI refresh (every 40 TTIs)        HBM -> LDS, 16 B per lane, transposed on the way in
 *   P1  PF EWMA update per user            ref: src/flows/radio-bearer.cpp:139-164
 *   P2  slice quotas (one wave, lanes = slices, beside P3)   ref: downlink-transport-scheduler.cpp:463-521
 *   P3  best user per (RBG, slice)         ref: :530-567   exact two-stage arg-max: FP32 ranking of 8 users
 *                                          per load, IEEE FP64 division only for the survivors
 *   P4  inter-slice assignment             ref: :249-272 GreedyByRow / :351-376 MaximizeCell = exact
 *                                          std::sort emulation (level-synchronous introsort loop + stable
 *                                          counting sort, rs_sort_device.h) + greedy scan; also :223-246 UpperBound
 *                                          (S segmented sorts), :378-451 VogelApproximate, and the NVS non-greedy
 *                                          sampler (downlink-nvs-scheduler.cpp:405-528) in place of P3/P4
 *   P5  apply + EESM link adaptation + DoStopSchedule counters     ref: :589-674, :170-221
 *
 * Template parameters of the cell body: SCHED = the reference's CLI scheduler number (1, 7, 8, 9, 10, 11; 101 = SubOpt, 103 = Vogel),
 * EPT = sort positions per thread (0: state in LDS, any size), FIXED = shape-specialised build, DIRECT = the drop-in
 * entry point's one-TTI form on caller-provided state.  Wave-level building blocks live in rs_wave.h.
 *
 * The same source is compiled twice: into the library with the cell shape as launch arguments, and at
 * run time (hiprtc, rs_jit.cpp) with the shape as compile-time constants (RS_JIT_*).
 *
 * No MFMA: the only matrix-shaped object (metric[R][U]) is consumed by an argmax.  All floating
 * point is IEEE FP64 add/mul/div in the reference's operation order; the file MUST be compiled with
 * -ffp-contract=off (the reference is x86-64 SSE2 without FMA).  libm never runs on the device: the
 * host evaluates the transcendental tables (rs_link_tables) and the device only compares.
 *
 * `ref:` paths are relative to /root/reference/src/protocolStack/mac/packet-scheduler/ unless they
 * start with src/.
 */
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "rs_device.h"
#include "rs_sort_emul.h"
#include "rs_wave.h"
#include "rs_sort_device.h"
#include "rs_interslice.h"

namespace {

#ifndef RS_P3_BLOCK_TOP
#define RS_P3_BLOCK_TOP 0 /* users per stage-1 block of the scan at the top of the TTI (0: the same as in the serial phase) */
#endif
template <int N> struct RsInt { static constexpr int v = N; };

/* x / 1000.0 correctly rounded in three instructions instead of the ~35 of the FP64 division sequence (ref: averageRate /= 1000.0,
 * downlink-transport-scheduler.cpp:685-689; one per user per TTI).  Markstein's theorem: if q is a faithful approximation of a / b
 * and y approximates 1 / b with a relative error below 2^-53, then r = fma(-b, q, a) is exact and fma(r, y, q) = RN(a / b).
 * Here y = 0.001 as a double (relative error 2.08e-17 = 2^-55.4), so q = RN(x * y) lies within 0.5 + 0.19 ulp of x / 1000, i.e.
 * it is one of its two neighbours.  No underflow: x >= 1.  tests/test_abi.py checks the identity in exact rational arithmetic
 * (random, near-midpoint and small-integer quotients). */
__device__ __forceinline__ double rs_div_1000(double x) {
  const double q = x * 0.001;
  const double r = __builtin_fma(-q, 1000.0, x);
  return __builtin_fma(r, 0.001, q);
}
#define RS_SPEC_NAP 2    /* s_sleep argument (x 64 cycles) while the scanning waves wait for the allocation */
#define RS_SERIAL_PRIO 3 /* issue priority of the wave that runs the serial end of the TTI (inter-slice policy, link adaptation) */
#define RS_SPEC_PRIO 0   /* issue priority of the scanning waves during the serial phase */
#ifndef RS_P3_BLOCK
#define RS_P3_BLOCK 32 /* users ranked per stage-1 block (multiple of 8, <= 32) */
#endif

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
#ifndef RS_STAMPS_W1_TID
#define RS_STAMPS_W1_TID 64
#endif
#if defined(RS_STAMPS) && defined(RS_STAMPS_W1)
/* second diagnostic clock: the first scanning thread (wave 1) during the serial phase, into the sort's sub-stamp slots */
#define RS_STAMP1(i)                                               \
  do {                                                             \
    if (tid == RS_STAMPS_W1_TID) {                                 \
      unsigned long long now_ = __builtin_readcyclecounter();      \
      sort_sub[i] += now_ - stamp1_prev;                           \
      stamp1_prev = now_;                                          \
    }                                                              \
  } while (0)
#else
#define RS_STAMP1(i) do { } while (0)
#endif

/*
 * The whole per-cell TTI loop.  FIXED = false: the cell shape comes from the launch arguments (the
 * kernels built into the library).  FIXED = true: shape, block size and the LDS carve are the compile-
 * time constants RS_JIT_* -- the form rs_jit.cpp compiles with hiprtc for one batch's exact shape
 * (constant divisors, constant LDS offsets, static LDS, fewer live scalars).
 */
#ifndef RS_JIT_S
#define RS_JIT_S 1
#define RS_JIT_U 1
#define RS_JIT_R 1
#define RS_JIT_G 1
#define RS_JIT_NT 64
#define RS_JIT_SCHED 8
#endif
#ifndef RS_JIT_WIN
#define RS_JIT_WIN 0 /* shape-specialised build: the longest 8-aligned slice window of the batch (0: not known at compile time) */
#endif
#ifndef RS_JIT_CARVEQ
#define RS_JIT_CARVEQ 0 /* shape-specialised build: rs_carve's `queue` argument as the host passes it (0; 1: gate scratch of a drop-in PF / NVS
                         * context; 2: queue model, bearers' hot words in LDS when they fit; 3: queue model, words in HBM) */
#endif

template <int SCHED, int EPT, bool FIXED, bool DIRECT, bool QUEUE = false>
__device__ __forceinline__ void rs_cell_body(const RsLaunch& p, unsigned char* lds) {
  static_assert(!QUEUE || (!DIRECT && (SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103 || SCHED == 1 || SCHED == 7)),
                "finite queues: batches of schedulers 1, 7, 8, 9, 101, 103");
  /* schedulers 1 and 7 with queues allocate RBG by RBG on wave 0 (the satisfied-flow break / the m_requiredRBs gate bind) */
  constexpr bool kQSerial = QUEUE && (SCHED == 1 || SCHED == 7);
  const int cell = blockIdx.x;
  /* only the drop-in entry point (DIRECT: one TTI on caller-provided state) uses these; batches never do, and their
   * kernels carry neither the code nor the registers */
  constexpr bool kDirect = DIRECT;
  /* per-PRB CQI for the link adaptation (reports that differ inside an RBG): the caller's block (drop-in mode) or the batch's
   * per-PRB epoch grids / trace rows; the metric always reads the RBG's first PRB, which is what the LDS grid holds */
  const bool per_prb = DIRECT ? p.prb_cqi != nullptr : (p.epochs_prb != nullptr || p.trace_prb != nullptr);
  const int queue_mode_in = DIRECT ? p.queue_mode : (QUEUE ? 1 : 0);
  /* customised-slice inputs per user: the caller's arrays (drop-in mode) or this cell's rows of the queue model's
   * (bit 0: the prioritized bearer has data; queue model only, bit 1: the user has any queued data = is in UsersToSchedule) */
  /* The queue model's per-bearer words (RS_QSTATE_BYTES_PER_USER per user) stay in LDS for the whole launch when the carve has
   * room (q_lds; a compile-time fact in a shape-specialised build, so its pointers are plain LDS pointers), else in HBM. */
  constexpr RsCarve kCvQ = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ);
  const bool q_lds = QUEUE && (FIXED ? kCvQ.q_lds != 0 : p.q_lds != 0);
  const int qU = FIXED ? RS_JIT_U : p.U;
  unsigned char* const qs = lds + (FIXED ? kCvQ.off_qstate : p.off_qstate);
  double* const qs_avg = (double*)qs;                  /* [2][U] m_averageTransmissionRate */
  double* const qs_next = qs_avg + 2 * qU;             /* [2][U] time stamp of the next arrival burst (+inf: none left) */
  double* const qs_headt = qs_next + 2 * qU;           /* [2][U] time stamp of the head burst (valid while the queue holds bytes) */
  double* const qs_hol = qs_headt + 2 * qU;            /* [U] head-of-line delay of the slice-priority bearer */
  int32_t* const qs_i = (int32_t*)(qs_hol + qU);       /* [7][2][U] head, tail, pk, frag, bytes, pkts, tx */
  long long* const qs_a0 = (long long*)(qs_i + 14 * qU); /* [2][U] first arrival burst of the bearer in the arr_* arrays */
  int32_t* const qs_narr = (int32_t*)(qs_a0 + 2 * qU); /* [2][U] number of its bursts */
  int32_t* const qs_hnf = qs_narr + 2 * qU;            /* [2][U] full packets of the head burst ... */
  int32_t* const qs_hla = qs_hnf + 2 * qU;             /* [2][U] ... and bytes of its last packet (valid while the queue holds packets) */
  uint8_t* const qs_kind = (uint8_t*)(qs_hla + 2 * qU); /* [U][2] */
  uint8_t* const qs_flags = qs_kind + 2 * qU;          /* [U] bit 0: prioritized bearer has data, bit 1: user has queued data */
  uint8_t* const qs_slice = qs_flags + qU;             /* [U] */
  const uint8_t* const prio_in = QUEUE ? (q_lds ? qs_flags : p.q_flags + (size_t)blockIdx.x * qU) : (DIRECT ? p.prio : nullptr);
  const double* const hol_in = QUEUE ? (q_lds ? qs_hol : p.q_hol + (size_t)blockIdx.x * qU) : (DIRECT ? p.hol : nullptr);
  const int tid = threadIdx.x;
  const int nt = FIXED ? RS_JIT_NT : (int)blockDim.x;
  const int lane = lane_id(), wave = wave_id(), nwaves = nt >> 6;
  /* (a shape-specialised drop-in kernel: RS_JIT_U is the context's user capacity -- it fixes the LDS carve -- while the users of one
   * call, their grid stride and, per-flow PF, their segments are launch arguments) */
  const int S = FIXED ? RS_JIT_S : p.S, U = (FIXED && !DIRECT) ? RS_JIT_U : p.U, R = FIXED ? RS_JIT_R : p.R, G = FIXED ? RS_JIT_G : p.G;
  constexpr RsCarve kCv = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ);
  /* byte offsets of the LDS arrays: constants in a shape-specialised build */
  struct Offs { int avgk, rcp, tab, slice, tx, misc, tbs, elems, sorted, items, sortx, cqi, queue, Upad, n_seg, n_items; };
  const Offs o = FIXED ? Offs{kCv.off_avgk, kCv.off_rcp, kCv.off_tab, kCv.off_slice, kCv.off_tx, kCv.off_misc, kCv.off_tbs,
                              kCv.off_elems, kCv.off_sorted, kCv.off_items, kCv.off_sortx, kCv.off_cqi, kCv.off_queue,
                              DIRECT ? p.Upad : kCv.Upad, DIRECT ? p.n_seg : kCv.n_seg, DIRECT ? p.n_items : kCv.n_items}
                       : Offs{p.off_avgk, p.off_rcp, p.off_tab, p.off_slice, p.off_tx, p.off_misc, p.off_tbs, p.off_elems,
                              p.off_sorted, p.off_items, p.off_sortx, p.off_cqi, p.off_queue, p.Upad, p.n_seg, p.n_items};
  constexpr bool kTransport = (SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103 || SCHED == 10);
  const int quota_wave = nwaves - 1; /* P2 runs on the last wave, beside the other waves' P3 */
  /* Speculative next-TTI metric scan.  In the schedulers whose inter-slice step and link adaptation run on
   * wave 0 alone, the other waves use that time to prepare TTI t+1 as if nobody were served in TTI t: the EWMA decay of every
   * user ((1 - beta) * avg, no bytes) and the best user of every (RBG, slice).  Serving a user can only LOWER its metric, so
   * an item whose speculative winner was not served keeps that winner exactly (first-maximum rule included); the few items
   * whose winner was served (~13 %) are listed and rescanned with the true averages after the TTI's closing barrier. */
  /* Measured on MI355X (profiles/r02_spec_notes.md): it pays while a slice's scan is short -- +1 % (sched 9) ... +7 % (sched 8) at
   * 25 UEs per slice, -4 ... -7 % at 50, where the rescans after the barrier cost more than the scan they replace: on up to
   * 32 UEs per slice on average; a shape-specialised build for a larger shape does not carry the code at all. */
  /* MaximizeCell's greedy scan (rs_interslice.h) in its vector form up to 32 RBGs, one record per step above (measured, 512
   * cells: 25 RBGs 31.1 against 30.3 M TTIs/s, 64 RBGs 12.4 against 13.1 M -- with many RBGs the scanning waves' LDS traffic slows
   * the vector form's atomics and compaction more than the serial loop's lane reads).  The vector form shortens the serial
   * phase to the point where the speculation no longer pays for its fix-up pass and its flags (31.1 M without, 30.8 M with the
   * averages alone prepared, 29.9 M with the scan): MaximizeCell speculates only where it keeps the serial scan.
   * (64 RBGs again in round 4, vector form with nothing speculated: 12.3 against 13.5 M -- profiles/r04_r64.md.) */
  constexpr int kVecMaxR = 32;
  constexpr bool kVecScan = SCHED == 9 && (!FIXED || RS_JIT_R <= kVecMaxR);
  const bool vec_scan = kVecScan && R <= kVecMaxR;
  /* Held winners (round 3): the winner of a (slice, RBG) item is NOT looked for again in a TTI in which it cannot
   * have changed -- the winner was not served in the previous TTI, the CQI grid is the same, and at the item's last scan its
   * stage-1 value led the slice by a margin that 40 TTIs of rounding and of the "+1" in (1 + avg) cannot use up.  Such TTIs scan
   * only the listed items (winner served, or margin too small), four lanes per item.  Supersedes the
   * speculative scan; -DRS_NO_HOLD restores round 2's behaviour.
   * Measured (512 cells, same box, against -DRS_NO_HOLD): 25 RBGs sched 9 32.8 against 31.9 M TTIs/s, sched 8 89.6 against 85.6,
   * sched 8 at 50 UEs per slice 74.5 against 62.2; 64 RBGs sched 8 37.3 against 48.9, sched 9 12.9 against 13.5 -- with many
   * RBGs the speculative scan hides in a long serial phase while 1 280 items are 2.5 chunks of listing per wave.  So: shape-
   * specialised builds of up to 32 RBGs (-DRS_HOLD_ALWAYS: every shape-specialised build); the built-in kernels, whose shape is
   * a run-time value, keep round 2's scan. */
#if defined(RS_NO_HOLD)
  constexpr bool kHoldSched = false;
#elif defined(RS_HOLD_ALWAYS)
  constexpr bool kHoldSched = FIXED && RS_JIT_NT <= 512 && !DIRECT && !QUEUE && (SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103);
#else
  /* (MaximizeCell's kernel carries the sort and is register-bound: it holds winners only when the host passed the batch's longest
   * slice window, so that the listed items' scan keeps 8 products per lane up to 32-user windows -- 16 above: 30.2 against 29.3 M
   * TTIs/s at 50 UEs per slice with the round-robin dealing, 28.3 against 29.4 with 64-item chunks) */
  constexpr bool kHoldSched = FIXED && RS_JIT_R <= 32 && RS_JIT_NT <= 512 && !DIRECT && !QUEUE &&
                              (SCHED == 8 || SCHED == 101 || SCHED == 103 || (SCHED == 9 && RS_JIT_WIN > 0 && RS_JIT_WIN <= 64));
#endif
#ifdef RS_NO_SPEC
  constexpr bool kSpecSched = false;
#else
  constexpr bool kSpecSched = !kHoldSched && !DIRECT && !QUEUE && (SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103) &&
                              (!FIXED || RS_JIT_U <= 32 * RS_JIT_S) && !(FIXED && kVecScan);
#endif
  const bool spec_enabled = kSpecSched && nwaves >= 2 && U <= 32 * S && !vec_scan;
  /* Schedulers 1 and 7 (round 4): their TTI ends with wave 0 alone (per-RBG reduction, link adaptation: 25-40 % of the TTI) while
   * the other waves idle.  Those waves prepare TTI t+1 meanwhile -- exactly, nothing speculative:
   *   both   the EWMA of every user as (1 - beta) * avg (what the reference computes for a user that was not served: + beta * 0
   *          adds nothing); wave 0 adds beta * rate for the users it served as soon as it knows their bytes -- the reference's sum
   *          of two rounded products (ref: src/flows/radio-bearer.cpp:139-164);
   *   NVS    SelectSliceToServe of TTI t+1 (downlink-nvs-scheduler.cpp:94-142 reads slice_ewma_time_ only, not the allocation),
   *          and, when that slice is NOT the one being served now (its users' averages then do not depend on this TTI's
   *          allocation) and TTI t+1 reads the same CQI grid, the whole metric scan of TTI t+1: such a TTI starts with its
   *          winners in place and is one workgroup barrier long.
   * Shape-specialised batch kernels without queues (the owners' byte counters live in registers there). */
#ifdef RS_NO_EARLY17
  constexpr bool kEarly17 = false;
#else
  /* (the per-flow PF scheduler gains only the EWMA: one multiply-add per user, worth moving once a thread owns several users --
   * same-box A/B, 512 cells: 1 000 UEs 96.0 against 91.5 M TTIs/s, 500 UEs 160.0 against 163.6; NVS: 184.5 against 149.8 M at
   * 500 UEs x 25 RBGs, 161.6 against 125.4 at 1 000 UEs, 167.1 against 125.7 at 64 RBGs; profiles/r04_sched17.md) */
  constexpr bool kEarly17 = FIXED && !DIRECT && !QUEUE && (SCHED == 7 || (SCHED == 1 && RS_JIT_U > RS_JIT_NT));
#endif

  double* s_avg = (double*)lds;
  double* s_avgk = (double*)(lds + o.avgk);
  float* s_rcp32 = (float*)(lds + o.rcp);
  int32_t* s_tx = (int32_t*)(lds + o.tx);
  double* s_num = (double*)(lds + o.tab); /* metric numerator per CQI */
  double* s_e = s_num + 16;
  double* s_x = s_e + 16;
  float* s_num32 = (float*)(s_x + 16);
  double* s_w = (double*)(lds + o.slice);
  double* s_sstate = s_w + 64;
  uint16_t* s_best_user = (uint16_t*)(lds + o.items); /* kSpecSched: two buffers of n_items, by TTI parity */
  double* s_best_metric = (double*)(lds + o.elems); /* sched 1 only (aliases elems) */
  uint32_t* s_elems = (uint32_t*)(lds + o.elems);
  uint32_t* s_sorted = (uint32_t*)(lds + o.sorted);
  Misc* m = (Misc*)(lds + o.misc);
  int32_t* s_tbs = (int32_t*)(lds + o.tbs); /* [R+1][16] TBS bits of n RBGs at a final CQI */
  uint8_t* s_cqi = lds + o.cqi; /* [R][Upad], Upad = 8 * odd >= U: conflict-free 8-byte column reads */
  const int Upad = o.Upad;

  const RsTables* tab = p.tab;
  RsCellScalars* scal = p.scal + cell;

  /* RadioBearer::m_cumulativeBytes / m_cumulativeRBs (ref: src/flows/radio-bearer.cpp:100-124).  A shape-specialised build
   * keeps them in registers of the thread that owns the user in P1: the serving lane of P5 leaves bytes | nPRB << 20 in
   * s_tx[u], the owner unpacks it when the next EWMA update consumes it, and the totals go to HBM once per launch (the
   * built-in kernels, whose users-per-thread count is a run-time value, add to HBM with fire-and-forget atomics instead).
   * Bit 30 of the stored tx word: "already in the HBM totals" (the last TTI's service, flushed with the launch). */
  constexpr bool kCumRegs = FIXED && !DIRECT && !QUEUE;
  constexpr int kKU = kCumRegs ? (RS_JIT_U + RS_JIT_NT - 1) / RS_JIT_NT : 1;
  /* 32-bit per launch: the host splits runs so that n_ttis * (largest transport block in bytes) < 2^31 (RS_MAX_TTIS_PER_LAUNCH) */
  int cum_b[kKU], cum_r[kKU];
#pragma unroll
  for (int k = 0; k < kKU; ++k) { cum_b[k] = 0; cum_r[k] = 0; }
  /* A shape-specialised drop-in kernel (rs_ctx_specialize) reads its per-call inputs -- averages, slice ids, the CQI grid -- from
   * the caller's pinned host block over PCIe.  The built-in kernel meets them one after the other (averages, slice ids twice
   * behind barriers, the grid in rounds of one 16-byte word per thread); here every thread issues all of its reads at kernel
   * entry and the phases below take them from registers.  Same-box A/B (us per rs_schedule_tti, tools/dropin_latency.cpp):
   * MaximizeCell 500 UEs x 25 RBGs 41.0 against 43.0, 100 x 64 53.8 against 55.0 -- but 500 x 64 (four grid words per thread)
   * 70.2 against 68.8, GreedyByRow 37.0 against 35.9, NVS 29.0 against 26.2: MaximizeCell with at most two grid words per thread. */
  constexpr int kPU = (FIXED && DIRECT) ? (RS_JIT_U + RS_JIT_NT - 1) / RS_JIT_NT : 1;
  constexpr int kPG = (FIXED && DIRECT) ? ((RS_JIT_U * RS_JIT_R + 15) / 16 + RS_JIT_NT - 1) / RS_JIT_NT : 1;
  constexpr bool kPrefetch = FIXED && DIRECT && SCHED == 9 && kPG <= 2;
  double pre_avg[kPU];
  int pre_sl[kPU];
  uint4 pre_grid[kPG];
  if constexpr (kPrefetch) {
    const uint4* src = (const uint4*)p.epochs;
    const int n16 = (int)(p.grid_stride >> 4);
#pragma unroll
    for (int k = 0; k < kPU; ++k) {
      const int u = tid + k * nt;
      pre_avg[k] = u < U ? p.avg[u] : 0.0;
      pre_sl[k] = u < U ? (int)p.user_slice[u] : 0;
    }
#pragma unroll
    for (int j = 0; j < kPG; ++j) {
      const int i = tid + j * nt;
      pre_grid[j] = i < n16 ? src[i] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
  /* ---------------- load the cell ---------------- */
#pragma unroll
  for (int ku = 0; ku < (kCumRegs ? kKU : 1); ++ku) {
    if (!kCumRegs) break;
    const int u = tid + ku * nt;
    if (u < U) {
      s_avg[u] = p.avg[(size_t)cell * U + u];
      int v = p.tx_bytes[(size_t)cell * U + u];
      if (v & RS_TX_COUNTED) { /* counted by the previous launch's flush: the consumption below must not count it again */
        cum_b[ku] -= v & RS_TX_BYTES_MASK;
        cum_r[ku] -= (v >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK;
        v &= ~RS_TX_COUNTED;
      }
      s_tx[u] = v;
    }
  }
  for (int u = tid; u < U && !kCumRegs; u += nt) {
    if (!kPrefetch) s_avg[u] = p.avg[(size_t)cell * U + u];
    s_tx[u] = p.tx_bytes[(size_t)cell * U + u];
    if (kDirect) { /* rs_schedule_tti: one row of per-user outputs, cleared here instead of by a memset */
      if (p.log_tbs) p.log_tbs[u] = 0;
      if (p.log_uinfo) p.log_uinfo[u] = 0;
    }
  }
  int16_t* const s_uoff = (int16_t*)(s_rcp32 + Upad + 16 * S); /* per user: window position minus user index (until then: the prefetched slice id) */
  if constexpr (kPrefetch) {
#pragma unroll
    for (int k = 0; k < kPU; ++k) {
      const int u = tid + k * nt;
      if (u < U) { s_avg[u] = pre_avg[k]; s_uoff[u] = (int16_t)pre_sl[k]; }
    }
  }
  /* TBS bits of n RBGs at a final CQI: the I_TBS step of CQI -> MCS -> I_TBS -> TBS folded in ([R+1][16]) */
  for (int i = tid; i < (R + 1) * 16; i += nt) s_tbs[i] = p.tbs_eff[(i >> 4) * 27 + tab->itbs_of_cqi[i & 15]];
  for (int i = tid; i < (R * Upad) >> 2; i += nt) ((uint32_t*)s_cqi)[i] = 0;
  for (int i = tid; i < Upad + 16 * S; i += nt) s_rcp32[i] = 0.0f;
  if (tid < 16) {
    s_num[tid] = (SCHED == 1 || SCHED == 11) ? tab->pfnum[tid] : tab->kbps[tid];
    s_e[tid] = tab->eesm_e[tid];
    s_x[tid] = tab->eesm_x[tid];
    s_num32[tid] = (float)((SCHED == 1 || SCHED == 11) ? tab->pfnum[tid] : tab->kbps[tid]);
    m->mcs_of_cqi[tid] = tab->mcs_of_cqi[tid];
    m->tbs1_of_cqi[tid] = tab->tbs1_syn[tid];
    m->ones16[tid] = 1.0f;
    m->eff16[tid] = tab->eff[tid];
  }
  if (tid < 2) { m->spec[tid].ctr_p1 = 0; m->spec[tid].ctr_p3 = 0; m->spec[tid].greedy_done = 0; m->spec[tid].n_fix = 0; }
  if (tid < S) {
    /* bit 0 = algo_epsilon, bit 1 = algo_psi, bit 2 = algo_alpha, bit 3 = algo_beta */
    m->eps_psi[tid] = (uint8_t)((p.eps[tid] ? 1 : 0) | (p.psi[tid] ? 2 : 0) | ((p.alpha && p.alpha[tid]) ? 4 : 0) | ((p.beta && p.beta[tid]) ? 8 : 0));
    s_w[tid] = p.weight[tid];
    s_sstate[tid] = p.slice_state[(size_t)cell * S + tid];
  }
  /* segments scanned in P3: slices (7/8/9) or fixed runs of RS_PF_SEG users (1) */
  if (SCHED == 1) {
    if (tid <= o.n_seg) m->seg_begin[tid] = min(tid * RS_PF_SEG, U);
  } else {
    if (tid <= S) m->seg_begin[tid] = U; /* filled below */
  }
  __syncthreads();
  if (SCHED != 1) {
    /* user_slice is non-decreasing: slice s = [first u with slice >= s, ...) */
    for (int u = tid; u < U; u += nt) {
      int s = kPrefetch ? (int)s_uoff[u] : (int)p.user_slice[u];
      int sp = u == 0 ? -1 : (kPrefetch ? (int)s_uoff[u - 1] : (int)p.user_slice[u - 1]);
      for (int q = sp + 1; q <= s; q++) m->seg_begin[q] = u;
    }
  }
  /* The stage-1 reciprocals of a slice live in their own 8-aligned window of s_rcp32, zero before the slice's first user
   * and after its last: the metric scan reads whole groups of 8 and a slot outside the slice multiplies to 0 without a
   * range test.  (Sched 1 has no slices: natural order, zeros behind the last user.) */
  if (SCHED != 1) {
    __syncthreads();
    if (wave == 0) {
      const int ub = lane < S ? m->seg_begin[lane] : 0, ue = lane < S ? m->seg_begin[lane + 1] : 0;
      const int wl = ue > ub ? ((ue + 7) & ~7) - (ub & ~7) : 0;
      const int wb = wave_scan_incl(wl) - wl;
      if (lane < S) m->rcp_off[lane] = wb - (ub & ~7);
    }
    __syncthreads();
    /* window offsets are multiples of 8 (>= 0): bit 0 carries the slice's algo_psi, so that P1 needs no other per-user table */
    if constexpr (kPrefetch) {
#pragma unroll
      for (int k = 0; k < kPU; ++k) {
        const int u = tid + k * nt;
        if (u < U) s_uoff[u] = (int16_t)(m->rcp_off[pre_sl[k]] | ((m->eps_psi[pre_sl[k]] & 2) ? 1 : 0));
      }
    } else {
      for (int u = tid; u < U; u += nt) s_uoff[u] = (int16_t)(m->rcp_off[p.user_slice[u]] | (p.psi[p.user_slice[u]] ? 1 : 0));
    }
  }
  double t = scal->t;
  double last_update = scal->last_update;
  long long last_sent = scal->last_sent;
  int reported = scal->reported;
  int cqi_row = scal->cqi_row;
  int served_prev = scal->served_prev;
  long long n_done = scal->n_done;
  WaveRng rng;
  rng.r = 0;
  rng.f = scal->rng_f;
  rng.b = scal->rng_b;
  if (wave == quota_wave && lane < 31) rng.r = scal->rng_r[lane];
  const int nb_rbs = R * G;
  int local_err = 0;
  __syncthreads();
  /* held winners (kHoldSched): one "held" bit per item (64 items per word, in the place of the second winner buffer), the users
   * served in the previous TTI (2 048-bit map) and the list of items to scan again in m->hist (free outside the counting sort) */
#ifndef RS_HOLD_MAX_AGE
#define RS_HOLD_MAX_AGE 40 /* TTIs a held winner is trusted without a full scan (the margin below is sized for it) */
#endif
  /* m->hist is shared by phases that never overlap: the counting sort (its histogram), then -- serial phase and top of the next
   * TTI -- the held winners' served map (hist[0..127]) and per-wave lists (64 entries per wave from hist[128]), the speculation's
   * served map + fix list, the queue model's per-slice words.  The lists must fit behind the map: */
  static_assert(!kHoldSched || 128 + 64 * (RS_JIT_NT / 64) <= (int)(sizeof(RsMisc::hist) / sizeof(uint16_t)),
                "held winners: the per-wave lists in RsMisc::hist are sized for at most 14 waves");
  unsigned long long* const hold_bits = (unsigned long long*)(lds + o.items + ((2 * o.n_items + 7) & ~7));
  uint32_t* const hold_served = (uint32_t*)m->hist; /* [64] */
  uint16_t* const hold_list = m->hist + 128;        /* 64 entries per wave: the items a wave scans again */
  int hold_win = 0; /* the longest 8-aligned slice window: 32 or 64 lanes per listed item (longer: no held winners) */
  if (kHoldSched) {
    int w = 0;
    if (lane < S && m->seg_begin[lane + 1] > m->seg_begin[lane]) w = ((m->seg_begin[lane + 1] + 7) & ~7) - (m->seg_begin[lane] & ~7);
    hold_win = wave_max(w);
  }
  /* (one word of held bits per wave and 64 * nwaves items, behind the winners: it must fit the second winner buffer's place) */
  const bool hold_ok = kHoldSched && hold_win > 0 && hold_win <= 64 && U <= 2048 &&
                       8 * nwaves * ((o.n_items + 64 * nwaves - 1) / (64 * nwaves)) + 8 <= 2 * o.n_items &&
                       (!(FIXED && RS_JIT_WIN > 0) || hold_win <= RS_JIT_WIN);
  int hold_age = 0;

#ifndef RS_STAMPS
  unsigned long long* sort_sub = nullptr;
#endif
#ifdef RS_STAMPS
  unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long sort_sub_store[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long* sort_sub = sort_sub_store;
  unsigned long long stamp_prev = __builtin_readcyclecounter();
  unsigned long long stamp1_prev = 0;
#endif
  /* position inside the CQI epoch and the epoch's index, kept as counters: a 64-bit modulo per TTI costs more than the
   * quota phase */
  long long epoch = 0;
  int epoch_pos = 0;
  if (!kDirect && p.cqi_mode == RS_CQI_EPOCHS) {
    epoch = n_done / p.refresh;
    epoch_pos = (int)(n_done - epoch * p.refresh);
    if (p.epoch_wrap) epoch %= p.n_epochs; /* the uploaded grids cycle (rs_batch_config.cqi_epoch_wrap) */
  }
  /* EESM decision thresholds X[1..13] as wave-uniform values for the whole launch (13 scalar register pairs when they fit) */
  double xthr_k[13];
  {
    const double xl = s_x[lane & 15];
#pragma unroll
    for (int k = 1; k <= 13; ++k)
      xthr_k[k - 1] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xl), k), __builtin_amdgcn_readlane(__double2loint(xl), k));
  }
  /* queue model scratch in LDS: m->hist is free outside the counting sort (and the speculation, which the queue model does not use) */
  /* schedulers 1 and 7: per-bearer scratch (rs_carve_with) */
  int32_t* const q_grant1 = (int32_t*)(lds + o.queue);   /* [U] sched 1: this TTI's grant of the user's second flow (the first: s_tx) */
  int32_t* const q_data0 = q_grant1 + U;                 /* [U] m_dataToTransmit of bearer 0 (0: no packets) */
  int32_t* const q_data1 = q_data0 + U;                  /* [U] ... of bearer 1 */
  int32_t* const q_need = q_data1 + U;                   /* [U] sched 7: m_requiredRBs minus the PRBs allocated so far */
  uint8_t* const q_done = (uint8_t*)(q_need + U);        /* [2U] sched 1: the flow is satisfied */
  if constexpr (kQSerial) {
    for (int u = tid; u < U; u += nt) q_grant1[u] = 0; /* (a launch ends with every grant consumed) */
    __syncthreads();
  }
  int32_t* const q_slice_prio = (int32_t*)m->hist;      /* [64] highest bearer priority with packets, per slice */
  int32_t* const q_slice_act = (int32_t*)m->hist + 64;  /* [64] the slice has a user with queued data */
  int32_t* const q_any = &m->nvs_slice; /* slices with data this TTI (0: RBsAllocation does not run); read in the serial phase,
                                         * i.e. after the counting sort reused m->hist: its own word (unused by these schedulers) */
    /* ---------------- finite queues: the bearers' MAC queues, one thread per user ----------------
   * ref: src/flows/MacQueue.cpp:86-200, src/protocolStack/rlc/um-rlc-entity.cpp:126-196, src/flows/radio-bearer.cpp:281-367,
   * downlink-transport-scheduler.cpp:105-221, packet-scheduler.cpp:305-335.  Per bearer (index = priority) the queue is a
   * window [head, tail) of its uploaded arrival bursts plus the progress inside the head burst; a burst = n_full packets of
   * RS_FULL_PACKET bytes and one last packet.  Order per TTI: DoStopSchedule of the previous TTI (grant split from the highest
   * priority down, RLC dequeue with 8 bytes of overhead per packet, fragmenting the last), EWMA of every bearer, the arrivals
   * up to now, then the user's record (bearers with packets, dataToTransmit, slice priority, head-of-line delay). */
  auto bearer_index = [&](int u, int b) -> size_t { return ((size_t)cell * 2 + b) * U + u; };
  /* the seven 32-bit words of bearer (b, u) -- HBM: seven arrays [cells][2][U] one after the other (RsLaunch::q_head ... b_tx);
   * LDS: [7][2][U] -- and its average */
  enum { QF_HEAD = 0, QF_TAIL = 1, QF_PK = 2, QF_FRAG = 3, QF_BYTES = 4, QF_PKTS = 5, QF_TX = 6 };
  int32_t* const qi_base = !QUEUE ? nullptr : (q_lds ? qs_i : p.q_head + (size_t)cell * 2 * U);
  const size_t qi_stride = !QUEUE ? 0 : (q_lds ? (size_t)2 * U : (size_t)p.n_cells * 2 * U);
  double* const qavg_base = !QUEUE ? nullptr : (q_lds ? qs_avg : p.b_avg + (size_t)cell * 2 * U);
  const uint8_t* const kind_of = !QUEUE ? nullptr : (q_lds ? qs_kind : p.bearer_kind);   /* [U][2] */
  const uint8_t* const slice_of = (QUEUE && q_lds) ? qs_slice : p.user_slice;
  uint8_t* const q_fl = !QUEUE ? nullptr : (q_lds ? qs_flags : p.q_flags + (size_t)cell * U);
  double* const q_ho = !QUEUE ? nullptr : (q_lds ? qs_hol : p.q_hol + (size_t)cell * U);
  auto QI = [&](int f, int b, int u) -> int32_t& { return qi_base[(size_t)f * qi_stride + b * U + u]; };
  auto queue_data = [&](int kind, int b, int u) -> int { /* m_dataToTransmit of a bearer with packets, else 0 */
    if (kind == 1) return 100000000;
    if (kind != 2) return 0;
    const int pk = QI(QF_PKTS, b, u);
    return pk > 0 ? QI(QF_BYTES, b, u) + 8 * pk : 0; /* GetQueueSizeWithMACHoverhead */
  };
  /* RadioBearer::m_cumulativeBytes / m_cumulativeRBs stay in HBM: touched for served bearers only, fire-and-forget adds */
  auto cum_add = [&](int64_t* w, long long v) { (void)atomicAdd((unsigned long long*)w, (unsigned long long)v); };
  /* one bearer's RLC dequeue of `sent` bytes (TransmissionProcedure): whole packets cost their data + 8 bytes, the last one may
   * leave as a fragment */
  auto rlc_dequeue = [&](int u, int b, int sent) {
    int left = sent, head = QI(QF_HEAD, b, u), pk = QI(QF_PK, b, u), frag = QI(QF_FRAG, b, u), qb = QI(QF_BYTES, b, u), qp = QI(QF_PKTS, b, u);
    const int head0 = head;
    const size_t a0 = q_lds ? (size_t)qs_a0[b * U + u] : (size_t)p.arr_off[(size_t)(cell * U + u) * 2 + b];
    /* the head burst's shape: cached beside the queue words (LDS) or read where it lies */
    int nfull = 0, last = 0;
    if (left > 8 && qp > 0) {
      nfull = q_lds ? qs_hnf[b * U + u] : p.arr_nfull[a0 + head];
      last = q_lds ? qs_hla[b * U + u] : p.arr_last[a0 + head];
    }
    while (left > 8 && qp > 0) {
      const int size_cur = pk < nfull ? RS_FULL_PACKET : last;
      const int data_cur = size_cur - frag;
      if (data_cur + 8 > left) { /* fragment */
        frag += left - 8;
        qb -= left - 8;
        left = 0;
        break;
      }
      left -= data_cur + 8;
      qb -= data_cur;
      qp -= 1;
      frag = 0;
      pk += 1;
      if (pk < nfull) { /* a run of untouched full packets leaves in one step */
        int k = left / (RS_FULL_PACKET + 8);
        k = k < nfull - pk ? k : nfull - pk;
        left -= k * (RS_FULL_PACKET + 8);
        qb -= k * RS_FULL_PACKET;
        qp -= k;
        pk += k;
      }
      if (pk >= nfull + (last > 0 ? 1 : 0)) {
        head += 1;
        pk = 0;
        if (qp > 0) { nfull = p.arr_nfull[a0 + head]; last = p.arr_last[a0 + head]; } /* (packets left: the next burst exists) */
      }
    }
    QI(QF_HEAD, b, u) = head; QI(QF_PK, b, u) = pk; QI(QF_FRAG, b, u) = frag; QI(QF_BYTES, b, u) = qb; QI(QF_PKTS, b, u) = qp;
    if (q_lds && head != head0 && qp > 0) {
      qs_headt[b * U + u] = p.arr_time[a0 + head];
      qs_hnf[b * U + u] = nfull;
      qs_hla[b * U + u] = last;
    }
  };
  auto stop_schedule_user = [&](int u) {
    if constexpr (SCHED == 1) {
      /* DL_PF_PacketScheduler::DoStopSchedule (dl-pf-packet-scheduler.cpp:60-125): every flow is credited its own transport
       * block in full and hands it to its RLC */
      long long user_bytes = 0, user_rbs = 0;
      for (int b = 0; b < 2; ++b) {
        int32_t* slot = b == 0 ? &s_tx[u] : &q_grant1[u];
        const int grant = *slot;
        if (grant == 0) continue;
        *slot = 0;
        const int bytes = grant & RS_TX_BYTES_MASK, nprb = (grant >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK;
        const size_t bi = bearer_index(u, b);
        QI(QF_TX, b, u) += bytes;
        cum_add(&p.b_cumb[bi], bytes);
        cum_add(&p.b_cumr[bi], nprb);
        user_bytes += bytes;
        user_rbs += nprb;
        if (kind_of[u * 2 + b] == 2) rlc_dequeue(u, b, bytes);
      }
      if (user_bytes) {
        cum_add(&p.cum_bytes[(size_t)cell * U + u], user_bytes);
        cum_add(&p.cum_rbs[(size_t)cell * U + u], user_rbs);
      }
      return;
    }
    const int grant = s_tx[u];
    if (grant == 0) return;
    s_tx[u] = 0;
    int avail = grant & RS_TX_BYTES_MASK;
    const int nprb = (grant >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK;
    long long user_bytes = 0;
    for (int b = 1; b >= 0 && avail > 0; --b) {
      const int kind = kind_of[u * 2 + b];
      const size_t bi = bearer_index(u, b);
      const int data = queue_data(kind, b, u);
      if (data <= 0) continue;
      const int sent = avail < data ? avail : data;
      avail -= sent;
      QI(QF_TX, b, u) += sent;
      cum_add(&p.b_cumb[bi], sent);
      cum_add(&p.b_cumr[bi], nprb);
      user_bytes += sent;
      if (kind == 2) rlc_dequeue(u, b, sent);
    }
    if (user_bytes) {
      cum_add(&p.cum_bytes[(size_t)cell * U + u], user_bytes); /* per-user totals for rs_batch_slice_bytes / read_state */
      cum_add(&p.cum_rbs[(size_t)cell * U + u], nprb);
    }
  };
  if constexpr (QUEUE) {
    if (q_lds) {
      /* load the bearers of this cell: every thread the users it owns in P1 (the only writer of these words) */
      const double kNever = __builtin_inf();
      for (int u = tid; u < U; u += nt) {
        for (int b = 0; b < 2; ++b) {
          const size_t bi = bearer_index(u, b);
          const int kind = p.bearer_kind[u * 2 + b];
          qs_kind[u * 2 + b] = (uint8_t)kind;
          qs_avg[b * U + u] = p.b_avg[bi];
          const size_t n = (size_t)p.n_cells * 2 * U;
#pragma unroll
          for (int f = 0; f < 7; ++f) qs_i[(f * 2 + b) * U + u] = p.q_head[(size_t)f * n + bi];
          double next = kNever, headt = 0.0;
          long long a0 = 0;
          int n_arr = 0, hnf = 0, hla = 0;
          if (kind == 2) {
            a0 = (long long)p.arr_off[(size_t)(cell * U + u) * 2 + b];
            n_arr = (int)((long long)p.arr_off[(size_t)(cell * U + u) * 2 + b + 1] - a0);
            const int tail = p.q_tail[bi];
            if (tail < n_arr) next = p.arr_time[a0 + tail];
            if (p.q_pkts[bi] != 0) {
              const int head = p.q_head[bi];
              headt = p.arr_time[a0 + head];
              hnf = p.arr_nfull[a0 + head];
              hla = p.arr_last[a0 + head];
            }
          }
          qs_next[b * U + u] = next;
          qs_headt[b * U + u] = headt;
          qs_a0[b * U + u] = a0;
          qs_narr[b * U + u] = n_arr;
          qs_hnf[b * U + u] = hnf;
          qs_hla[b * U + u] = hla;
        }
        qs_slice[u] = p.user_slice[u];
        qs_flags[u] = 0;
        qs_hol[u] = 0.0;
      }
      __syncthreads();
    }
  }
  bool have_spec = false; /* this TTI's EWMA, metric scan and quotas were prepared during the previous TTI's serial phase */
  bool have_quota = false; /* held winners: this TTI's quotas were worked out by the quota wave during the previous TTI's serial phase */
  bool have_ewma = false;  /* held winners: ... and so were the PF averages and terms (decay by the idle waves, served users by wave 0) */
  int pre_listed = -1;     /* held winners: this wave's list for this TTI was packed during the previous TTI's serial phase (-1: no) */
  bool have_scan = false;  /* NVS (kEarly17): this TTI's winners were found during the previous TTI's serial phase */
  for (int tti = 0; tti < p.n_ttis; ++tti) {
    RS_STAMP(11);
    auto prb_ptr = [&](int user, int r2) -> const uint8_t* { /* the G PRBs of RBG r2 as `user` reported them */
      if (DIRECT) return p.prb_cqi + ((size_t)user * R + r2) * G;
      if (p.cqi_mode == RS_CQI_EPOCHS) {
        const long long e = epoch < p.n_epochs ? epoch : (long long)p.n_epochs - 1;
        return p.epochs_prb + ((size_t)cell * p.n_epochs + (size_t)e) * p.grid_stride_prb + ((size_t)user * R + r2) * G;
      }
      return p.trace_prb + (((size_t)p.user_trace[(size_t)cell * U + user] * p.n_rows + cqi_row) * R + r2) * G;
    };
    /* ---------------- P0: CQI refresh ---------------- */
    bool grid_loaded = false; /* this TTI scans a new CQI grid (held winners: everything is scanned again) */
    if (p.cqi_mode == RS_CQI_EPOCHS) {
      /* a launch that starts inside an epoch loads that epoch's grid first: LDS does not survive between launches */
      if (kDirect || tti == 0 || epoch_pos == 0) {
        grid_loaded = true;
        long long e = kDirect ? 0 : epoch;
        if (e >= p.n_epochs) { local_err = RS_CQI_EPOCHS; e = p.n_epochs - 1; }
        /* HBM grid is [U][R] (one row per UE, like the reference's per-UE CQI vectors); LDS keeps it
         * RBG-major [R][Upad] so that the metric scan reads 8 consecutive UEs of one RBG per load */
        const uint4* src = (const uint4*)(p.epochs + ((size_t)cell * p.n_epochs + (size_t)e) * p.grid_stride);
        const int n16 = (int)(p.grid_stride >> 4);
        const int total = U * R;
        auto scatter16 = [&](int i, const uint4 w) { /* 16 grid bytes [u][r] -> the RBG-major LDS grid */
          const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
          int idx = i << 4;
          int u = idx / R, r = idx - u * R;
#pragma unroll
          for (int k = 0; k < 16; ++k, ++idx) {
            if (idx < total) s_cqi[r * Upad + u] = (uint8_t)(ww[k >> 2] >> ((k & 3) * 8));
            if (++r == R) { r = 0; ++u; }
          }
        };
        if constexpr (kPrefetch) {
#pragma unroll
          for (int j = 0; j < kPG; ++j)
            if (tid + j * nt < n16) scatter16(tid + j * nt, pre_grid[j]);
        } else {
          for (int i = tid; i < n16; i += nt) scatter16(i, src[i]);
        }
      }
    } else if (p.cqi_mode == RS_CQI_TRACE) {
      /* ref: src/device/CqiManager/cqi-manager.cpp:105-123 (interval 40),
       *      src/protocolStack/mac/enb-mac-entity.cc:189-191 */
      bool load = tti == 0 && reported; /* a launch that starts between two reports reloads the last reported row */
      if (!reported || ((int)(t * 1000) - last_sent) >= 40) {
        reported = 1;
        last_sent = (long long)(t * 1000);
        int stamp = (int)(t * 1000 / 40);
        cqi_row = stamp % p.row_mod;
        if (cqi_row >= p.n_rows) { local_err = RS_CQI_TRACE; cqi_row = 0; }
        load = true;
      }
      if (load) {
        grid_loaded = true;
        for (int i = tid; i < U * R; i += nt) {
          int u = i / R, r = i - u * R;
          int tr = p.user_trace[(size_t)cell * U + u];
          s_cqi[r * Upad + u] = p.trace[((size_t)tr * p.n_rows + cqi_row) * R + r];
        }
      }
    }
    /* which buffers this TTI's winners and records live in (see kSpecSched) */
    const int n_items_rt = o.n_items;
    uint16_t* const cur_bu = s_best_user + ((kSpecSched && (tti & 1)) ? n_items_rt : 0);
    uint16_t* const nxt_bu = s_best_user + ((kSpecSched && !(tti & 1)) ? n_items_rt : 0);
    /* held winners: pack the items of chunk k0 that this wave has to scan again into its list (item = k * nwaves + wave: the
     * items a served user leads are neighbours -- one slice, many RBGs -- so every wave gets its share); returns their number */
    uint16_t* const hold_wl = hold_list + wave * 64;
    auto hold_pack = [&](int k0, int for_wave) -> int { /* for_wave: whose items (wave 1 also packs wave 0's in the serial phase) */
      const int it_l = (k0 + lane) * nwaves + for_wave;
      const bool in = it_l < n_items_rt;
      const int w = in ? (int)cur_bu[it_l] : 0xFFFF;
      const bool held_bit = ((hold_bits[(k0 >> 6) * nwaves + for_wave] >> lane) & 1ull) != 0ull;
      const int sg_l = FIXED ? it_l / RS_JIT_R : idiv_small(in ? it_l : 0, R);
      const bool psi_on = (m->eps_psi[in ? sg_l : 0] & 2) != 0;
      const bool was_served = w != 0xFFFF && ((hold_served[(w & 2047) >> 5] >> (w & 31)) & 1u) != 0u;
      const bool need = in && w != 0xFFFF && (!held_bit || (was_served && psi_on));
      const unsigned long long mk = __ballot(need);
      if (need) hold_list[for_wave * 64 + __popcll(mk & ((1ull << lane) - 1ull))] = (uint16_t)it_l;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      return __popcll(mk);
    };
    /* sched 9 scans s_sorted in the serial phase, so the next TTI's records can go straight to s_elems; the policies that read
     * the records themselves (8, 101, 103) alternate between s_elems and the otherwise unused s_sorted */
    constexpr bool kAltRec = kSpecSched && SCHED != 9;
    uint32_t* const cur_rec = (kAltRec && (tti & 1)) ? s_sorted : s_elems;
    uint32_t* const nxt_rec = (kAltRec && !(tti & 1)) ? s_sorted : s_elems;
    RsSpecFlags* const fl_cur = &m->spec[tti & 1];        /* flags of this TTI's serial phase */
    RsSpecFlags* const fl_prev = &m->spec[(tti & 1) ^ 1]; /* ... of the previous one (read by the fix-up below) */
    uint32_t* const served_bits = (uint32_t*)m->hist;     /* [32] users served in the serial phase (m->hist is free then) */
    uint16_t* const fix_list = m->hist + 64;              /* items to rescan, RS_FIX_CAP entries */
    /* items the scanning waves take in the serial phase: whole rounds of nt - 64 lanes (at least one round) */
    auto spec_items = [&](int n) {
      const int nsp_ = nt > 64 ? nt - 64 : 64; /* (one-wave cells never speculate) */
      /* the serial phase lasts ~330 cycles per RBG, a round of scans ~4 000-5 000: with many RBGs every round fits (64 RBGs:
       * 13.06 instead of 12.92 M TTIs/s), with 25 a second, nearly empty round would outlast the serial wave */
      const int rounds_all = (n + nsp_ - 1) / nsp_;
      if (rounds_all * 16 <= R) return n;
      const int rounds = n / nsp_;
      return rounds == 0 ? n : rounds * nsp_;
    };
    /* averageRate /= 1000.0 (ref: :685-689).  The three-instruction form needs an ordinary operand (no underflow, finite); the
     * drop-in entry point takes any double from its caller, and when one is out of range (p.exact_scan, set by the host) it
     * divides for real -- inf / 1000 is inf, the short form would give NaN. */
    auto div_1000 = [&](double k) -> double {
      if constexpr (DIRECT) {
        if (p.exact_scan) return k / 1000.0;
      }
      return rs_div_1000(k);
    };
    /* PF terms of a user whose average is `a`: exact denominator and stage-1 reciprocal (ref: :685-689) */
    auto pf_terms = [&](int u, double a) {
      double k = 1;
      k += a;
      k = div_1000(k);
      if constexpr (DIRECT) {
        if (p.gen_exp) k = a; /* general exponents: the caller's block already holds pow(avg_kbps, psi) (host libm) */
      }
      s_avgk[u] = k;
      const int uo = s_uoff[u];
      /* stage-1 ranking only, never part of a result; psi == 0 slices rank on the numerator */
      s_rcp32[u + (uo & ~7)] = (uo & 1) ? __builtin_amdgcn_rcpf((float)k) : 1.0f; /* v_rcp_f32, 1 ulp */
    };
    /* ---------------- P1: PF EWMA (ref: src/flows/radio-bearer.cpp:139-164) ---------------- */
    if constexpr (QUEUE) {
      if (tid < 64) { q_slice_prio[tid] = 0; q_slice_act[tid] = 0; }
      __syncthreads();
      const bool do_ewma = !(t == last_update);
      const double dt = t - last_update;
      for (int u = tid; u < U; u += nt) {
        stop_schedule_user(u);
        bool has[2] = {false, false};
        double bavg[2] = {0, 0};
        for (int b = 0; b < 2; ++b) {
          const int kind = kind_of[u * 2 + b];
          if (kind == 0) continue;
          /* RadioBearer::UpdateAverageTransmissionRate: every bearer, scheduled or not */
          double a = qavg_base[b * U + u];
          if (do_ewma) {
            const double rate = (double)(QI(QF_TX, b, u) * 8) / dt;
            const double beta = 0.02;
            a = ((1 - beta) * a) + (beta * rate);
            if (a < 1) a = 1;
            qavg_base[b * U + u] = a;
            QI(QF_TX, b, u) = 0;
          }
          bavg[b] = a;
          if (kind == 2) {
            /* the applications' Send() events with a time stamp up to now (MacQueue::Enqueue per packet).  With the state in LDS
             * the next burst's time stamp is cached: a TTI without arrivals touches no HBM */
            int qp = QI(QF_PKTS, b, u);
            if (!q_lds || qs_next[b * U + u] <= t) {
              const size_t a0 = q_lds ? (size_t)qs_a0[b * U + u] : (size_t)p.arr_off[(size_t)(cell * U + u) * 2 + b];
              const int n_arr = q_lds ? qs_narr[b * U + u] : (int)((size_t)p.arr_off[(size_t)(cell * U + u) * 2 + b + 1] - a0);
              int tail = QI(QF_TAIL, b, u), qb = QI(QF_BYTES, b, u);
              const int tail0 = tail, qp0 = qp;
              while (tail < n_arr && p.arr_time[a0 + tail] <= t) {
                const int nfull = p.arr_nfull[a0 + tail], last = p.arr_last[a0 + tail];
                qb += nfull * RS_FULL_PACKET + last;
                qp += nfull + (last > 0 ? 1 : 0);
                tail += 1;
              }
              if (tail != tail0) { QI(QF_TAIL, b, u) = tail; QI(QF_BYTES, b, u) = qb; QI(QF_PKTS, b, u) = qp; }
              if (q_lds) {
                qs_next[b * U + u] = tail < n_arr ? p.arr_time[a0 + tail] : __builtin_inf();
                /* an empty queue's head is its first new burst */
                if (qp0 == 0 && tail != tail0) {
                  const int head = QI(QF_HEAD, b, u);
                  qs_headt[b * U + u] = p.arr_time[a0 + head];
                  qs_hnf[b * U + u] = p.arr_nfull[a0 + head];
                  qs_hla[b * U + u] = p.arr_last[a0 + head];
                }
              }
            }
            has[b] = qp > 0;
          } else {
            has[b] = true; /* InfiniteBuffer: HasPackets() is always true */
          }
        }
        const bool active = has[0] || has[1];
        if constexpr (kQSerial) {
          q_data0[u] = has[0] ? queue_data(kind_of[u * 2], 0, u) : 0;
          q_data1[u] = has[1] ? queue_data(kind_of[u * 2 + 1], 1, u) : 0;
          q_done[2 * u] = 0;
          q_done[2 * u + 1] = 0;
        }
        if constexpr (SCHED == 1) {
          /* flows, not users: each bearer competes with its own average (dl-pf-packet-scheduler.cpp:128-140) */
          s_avg[u] = bavg[0];
          s_avgk[u] = bavg[1];
          continue;
        }
        double k = 1; /* averageRate = 1; += every bearer of the record, in index order (:681-686) */
        if (has[0]) k += bavg[0];
        if (has[1]) k += bavg[1];
        k = rs_div_1000(k);
        s_avgk[u] = k;
        if (active) {
          const int sl = slice_of[u];
          atomicMax(&q_slice_prio[sl], has[1] ? 1 : 0);
          q_slice_act[sl] = 1;
        }
        q_fl[u] = active ? 2 : 0;
      }
      last_update = t;
      __syncthreads();
      for (int u = tid; u < U && SCHED != 1; u += nt) {
        const int flags = q_fl[u];
        const int sl = slice_of[u];
        const int uo = s_uoff[u];
        float r32 = 0.0f;
        if (flags & 2) {
          r32 = (uo & 1) ? __builtin_amdgcn_rcpf((float)s_avgk[u]) : 1.0f;
          const int sl_bits = m->eps_psi[sl];
          if (sl_bits & 4) {
            /* customised slice (ref: :694-711): the slice's priority = the highest bearer priority with packets in the slice */
            const int pb = q_slice_prio[sl];
            const int kind = kind_of[u * 2 + pb];
            const bool has_data = queue_data(kind, pb, u) != 0;
            double hol = 0.0; /* GetHeadOfLinePacketDelay: 0 with an empty MAC queue (an InfiniteBuffer bearer has none) */
            if (kind == 2 && QI(QF_BYTES, pb, u) != 0) {
              double head_time;
              if (q_lds) {
                head_time = qs_headt[pb * U + u];
              } else {
                const size_t a0 = (size_t)p.arr_off[(size_t)(cell * U + u) * 2 + pb];
                head_time = p.arr_time[a0 + QI(QF_HEAD, pb, u)];
              }
              hol = t - head_time;
              if (hol < 0.00001) hol = 0.00001;
            }
            q_ho[u] = hol;
            q_fl[u] = (uint8_t)(2 | (has_data ? 1 : 0));
            r32 = !has_data ? 0.0f : ((SCHED == 7 || (sl_bits & 8) != 0) ? r32 * (float)hol : r32);
          }
        }
        s_rcp32[u + (uo & ~7)] = r32;
      }
      __syncthreads();
    } else
    if (!have_spec && !have_ewma) {
      const bool do_ewma = !kDirect && !(t == last_update);
      const double dt = t - last_update;
      auto ewma_user = [&](int u, int ku) {
        double a = s_avg[u];
        if ((kSpecSched || kHoldSched || kEarly17) && a < 1) a = 1; /* an update prepared in the serial phase leaves the unclamped product behind */
        if (do_ewma) {
          int txb = s_tx[u];
          if (kCumRegs) {
            cum_r[ku < kKU ? ku : 0] += (txb >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK;
            txb &= RS_TX_BYTES_MASK;
            cum_b[ku < kKU ? ku : 0] += txb;
          }
          double rate = (double)(txb * 8) / dt;
          const double beta = 0.02;
          a = ((1 - beta) * a) + (beta * rate);
          if (a < 1) a = 1;
          s_avg[u] = a;
          s_tx[u] = 0;
        }
        if (SCHED != 1) {
          if (!queue_mode_in) {
            pf_terms(u, a);
          } else {
            /* ref: :685-689  averageRate = 1 + sum(avg); averageRate /= 1000.0 */
            double k = 1;
            k += a;
            k = div_1000(k);
            if constexpr (DIRECT) {
              if (p.gen_exp) k = a;
            }
            s_avgk[u] = k;
            const int uo = s_uoff[u];
            float r32 = (uo & 1) ? __builtin_amdgcn_rcpf((float)k) : 1.0f;
            /* customised slice (ref: :694-711): metric 0 while the prioritized bearer is empty, times the
             * head-of-line delay when beta (sched 7: always) -- folded into the stage-1 factor */
            const int sl = kPrefetch ? pre_sl[ku < kPU ? ku : 0] : (int)p.user_slice[u];
            if (m->eps_psi[sl] & 4) { /* algo_alpha */
              const bool has = prio_in ? (prio_in[u] & 1) != 0 : true;
              const bool use_hol = SCHED == 7 || SCHED == 11 || (m->eps_psi[sl] & 8) != 0; /* algo_beta */
              r32 = !has ? 0.0f : (use_hol ? r32 * (float)hol_in[u] : r32);
            }
            s_rcp32[u + (uo & ~7)] = r32;
          }
        } else {
          s_rcp32[u] = __builtin_amdgcn_rcpf((float)a);
        }
      };
      if constexpr (kCumRegs) {
#pragma unroll
        for (int ku = 0; ku < kKU; ++ku)
          if (tid + ku * nt < U) ewma_user(tid + ku * nt, ku);
      } else if constexpr (kPrefetch) {
#pragma unroll
        for (int ku = 0; ku < kPU; ++ku)
          if (tid + ku * nt < U) ewma_user(tid + ku * nt, ku);
      } else {
        for (int u = tid, ku = 0; u < U; u += nt, ++ku) ewma_user(u, ku);
      }
      if (!kDirect) last_update = t;
      __syncthreads();
    } else {
      /* the previous serial phase has already updated every average for this TTI (speculatively, then exactly for the
       * served users); what is left is bookkeeping: the owners count the bytes granted in the previous TTI */
      if constexpr (kCumRegs) {
#pragma unroll
        for (int ku = 0; ku < kKU; ++ku) {
          const int u = tid + ku * nt;
          if (u < U) {
            int v = s_tx[u];
            if (v != 0) {
              /* (NVS with its winners in place runs this TTI without a barrier before wave 0's next grant: take the word atomically) */
              if (kEarly17 && SCHED == 7) v = atomicExch(&s_tx[u], 0);
              else s_tx[u] = 0;
              cum_r[ku] += (v >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK;
              cum_b[ku] += v & RS_TX_BYTES_MASK;
            }
          }
        }
      }
      last_update = t;
      /* (the barrier that ends the EWMA phase also separates the CQI refresh from the scans that read the grid: a TTI whose
       * averages were prepared beforehand still needs it when P0 loaded a grid -- round 2's speculation never prepared such a TTI) */
      if (grid_loaded) __syncthreads();
    }
    RS_STAMP(0);

    /* ---------------- P2: quotas / slice choice (one wave, lanes = slices) ---------------- */
    /* the quota phase of the transport schedulers in two halves: the draws and the two rotations (which slice receives
     * the remainders) depend on the rand() stream only; the targets need slice_rbs_offset_ of the previous TTI.  In a
     * speculated TTI the quota wave runs the first half while wave 0 is still deciding the allocation. */
    bool q_has = false;
    int q_nonempty = 0;
    bool q_first0 = false, q_first1 = false; /* my slice receives the remainder of the PRBs / of the RBGs */
    auto quota_draws = [&](int served_before) {
      /* ref: :463-521 */
      const bool in = lane < S;
      q_has = in && (QUEUE ? q_slice_act[lane] != 0 : (m->seg_begin[lane + 1] > m->seg_begin[lane]));
      q_nonempty = __popcll(__ballot(q_has));
      if (QUEUE && lane == 0) *q_any = q_nonempty;
      int r0 = p.rand0, r1 = p.rand1;
      if (!kDirect && !(QUEUE && q_nonempty == 0)) { /* (no user with queued data: RBsAllocation does not run, :160-165) */
        if (p.phy_draws)
          for (int i = 0; i < served_before; i++) (void)rng.next();
        r0 = rng.next();
        r1 = rng.next();
      }
      /* first non-empty slice in the rotation k = (i + rand) % S, i = 0..S-1 */
      const int r0m = (int)((unsigned)r0 % (unsigned)S), r1m = (int)((unsigned)r1 % (unsigned)S);
      int pos0 = lane - r0m;
      pos0 = pos0 < 0 ? pos0 + S : pos0;
      pos0 = q_has ? pos0 : 1 << 20;
      q_first0 = pos0 == wave_min(pos0);
      int pos1 = lane - r1m;
      pos1 = pos1 < 0 ? pos1 + S : pos1;
      pos1 = q_has ? pos1 : 1 << 20;
      q_first1 = pos1 == wave_min(pos1);
    };
    auto quota_targets = [&]() {
      const bool in = lane < S;
      int target = 0;
      if (q_has) target = (int)(nb_rbs * s_w[lane] + s_sstate[lane]);
      int extra = nb_rbs - wave_sum(target);
      if (QUEUE && q_nonempty == 0) { /* nothing to schedule this TTI */
        m->target[lane] = 0;
        m->quota[lane] = 0;
        return;
      }
      const int share = idiv_small(extra, q_nonempty), rem = extra - share * q_nonempty; /* C '/' and '%' */
      if (q_has) {
        target += share;
        if (q_first0) target += rem;
      }
      int quota = in ? idiv_small(target, G) : 0;
      int extra_g = R - wave_sum(quota);
      const int share_g = idiv_small(extra_g, q_nonempty), rem_g = extra_g - share_g * q_nonempty;
      if (q_has) {
        quota += share_g;
        if (q_first1) quota += rem_g;
      }
      m->target[lane] = target;
      m->quota[lane] = quota;
    };
    /* SelectSliceToServe (one wave, lanes = slices); also called one TTI ahead, during the previous TTI's serial phase (kEarly17) */
    auto nvs_pick = [&](int32_t* slice_out) {
        /* SelectSliceToServe, ref: downlink-nvs-scheduler.cpp:94-142 */
        int pick;
        if (kDirect) {
          pick = 0; /* the caller passes only the served slice's users */
        } else {
          const bool in = lane < S;
          /* slices with queued data (ref: :101-121; with queues: a bearer with packets and dataToTransmit > 0) */
          const bool has = in && (QUEUE ? q_slice_act[lane] != 0 : (m->seg_begin[lane + 1] > m->seg_begin[lane]));
          double ew = in ? s_sstate[lane] : 1.0;
          unsigned long long zero = __ballot(has && ew == 0);
          unsigned long long hasm = __ballot(has);
          /* a zero-ewma slice ends the scan and wins outright; otherwise '>=' keeps the LAST maximum */
          if (zero) {
            pick = __ffsll((long long)zero) - 1;
          } else {
            /* scores are >= 0 (-1 for a slice without users), so they order like the integer pair (high word signed, low
             * word unsigned): two DPP max reductions instead of a shuffle tree on doubles */
            const double score = has ? s_w[lane] / ew : -1.0;
            const int hi = __double2hiint(score);
            const int lo = (int)((unsigned)__double2loint(score) ^ 0x80000000u);
            const int mhi = wave_max(hi);
            const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
            const unsigned long long top = __ballot(hi == mhi && lo == mlo) & hasm;
            pick = top ? 63 - __clzll((long long)top) : 0;
          }
          const double beta = 0.01;
          if (has) {
            double e2 = (1 - beta) * ew;
            if (lane == pick) e2 += beta * 1;
            s_sstate[lane] = e2;
          }
        }
        if (lane == 0) *slice_out = pick;
        m->target[lane] = 0;
        m->quota[lane] = 0;
    };
    /* the served slice's word, by TTI parity when picks are made a TTI ahead: a TTI that starts without a barrier may still be
     * reading its own while the quota wave already writes the next one's */
    int32_t* const nvs_word_cur = (kEarly17 && (tti & 1)) ? &m->pad[0] : &m->nvs_slice;
    int32_t* const nvs_word_nxt = (kEarly17 && !(tti & 1)) ? &m->pad[0] : &m->nvs_slice;
    auto quota_phase = [&](int served_before) {
      if (!kTransport && !kDirect && p.phy_draws)
        for (int i = 0; i < served_before; i++) (void)rng.next();
      if (kTransport) {
        quota_draws(served_before);
        quota_targets();
      } else if (SCHED == 7 || SCHED == 11) {
        nvs_pick(nvs_word_cur);
      } else {
        m->target[lane] = 0;
        m->quota[lane] = 0;
      }
    };
    if (wave == quota_wave && !have_spec && !have_quota) quota_phase(served_prev);
    if (kEarly17 && SCHED == 7 && have_quota && wave == quota_wave && p.phy_draws) /* the pick was made a TTI ahead; the error model's */
      for (int i = 0; i < served_prev; i++) (void)rng.next();                       /* draws (one per user served) still come first   */
    int seg_lo = 0;   /* NVS: the served slice */
    int nvs_runs = 1; /* sched 7: runs of the served slice scanned in P3 */
    if (SCHED == 7 || SCHED == 11) {
      if (!(kEarly17 && have_quota)) __syncthreads(); /* P3 scans the slice P2 picked (a pick made a TTI ahead lies behind that TTI's closing barrier) */
      seg_lo = kDirect ? 0 : *nvs_word_cur;
    }
    const int seg_this = seg_lo; /* (the scanning waves of an NVS cell move seg_lo on to the next TTI's slice in the serial phase) */
    RS_STAMP(1);

    if constexpr (SCHED == 11) {
      /* ---------------- NVS non-greedy sampler, ref: downlink-nvs-scheduler.cpp:405-528 ----------------
       * RS_NVS_SAMPLES times every UE of the served slice draws a CQI index max(highest_cqi - rand() % 4, 1); per sample
       * every RBG goes to the first UE with the largest eff(index)*180000/(1+avg) among the UEs whose CQI on the RBG
       * reaches their index (metric 0 otherwise, strict '<' from -1), the sample scores the sum of the winners' metrics
       * in RBG order, and the first sample with the strictly largest score (from 0) is applied.  The 4 possible metrics
       * of a UE are computed once; samples run in batches: the generator wave draws a batch (31 ring words per step),
       * one thread per (sample, RBG) scans the slice, one lane per sample adds up, wave 0 keeps the best. */
      unsigned char* nv = lds + o.sortx;
      double* nv_val = (double*)nv;
      double* nv_hm = (double*)(nv + 32 * U);
      uint8_t* nv_draw = nv + 32 * U + 8 * RS_NVS_BATCH * R;
      uint16_t* nv_ha = (uint16_t*)(nv_draw + RS_NVS_DRAW_BYTES);
      uint8_t* nv_high = (uint8_t*)(nv_ha + RS_NVS_BATCH * R) + 128;
      int ub = m->seg_begin[seg_lo], ue = m->seg_begin[seg_lo + 1];
      if (kDirect) { ub = 0; ue = U; }
      const int n = ue - ub;
      for (int i = tid; i < n; i += nt) {
        const int u = ub + i;
        int h = 0;
        for (int r = 0; r < R; ++r) { /* :417-424 */
          const int c = s_cqi[r * Upad + u];
          h = c > h ? c : h;
        }
        nv_high[i] = (uint8_t)h;
        double rate = 1; /* UserToSchedule::GetAverageTransmissionRate, packet-scheduler.cpp:424-433 */
        rate += s_avg[u];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int mc = h - k > 1 ? h - k : 1;
          nv_val[i * 4 + k] = s_num[mc] / rate; /* sEff * 180000 / rate (:519-520) */
        }
      }
      if (tid < R) s_best_user[tid] = 0xFFFF;
      /* the generator wave draws batch b+1 into the other half of the draw buffer while the other waves scan batch b */
      const bool overlap = !kDirect && nwaves > 1;
      const int half = RS_NVS_DRAW_BYTES / 2;
      int bs = (overlap ? half : RS_NVS_DRAW_BYTES) / (n > 0 ? n : 1);
      bs = bs > RS_NVS_BATCH ? RS_NVS_BATCH : bs;
      auto draw_batch = [&](int b0, uint8_t* dst) { /* generator wave only */
        const int nbb = RS_NVS_SAMPLES - b0 < bs ? RS_NVS_SAMPLES - b0 : bs;
        const int total = nbb * n;
        int d0 = 0;
        if (total >= 31) { /* whole blocks stream in the chain-major layout (row-local DPP scans, no LDS round trips) */
          const int j = WaveRng::cm_index();
          rng.to_chain_major();
          for (; d0 + 31 <= total; d0 += 31) {
            const uint32_t x = rng.next_block_chain_major();
            if (j >= 0) dst[d0 + j] = (uint8_t)((x >> 1) & 3u); /* rand() % 4 */
          }
          rng.to_age_order();
        }
        if (d0 < total) {
          const uint32_t x = rng.next_block(total - d0);
          if (lane < total - d0) dst[d0 + lane] = (uint8_t)((x >> 1) & 3u);
        }
      };
      if (overlap && wave == quota_wave) draw_batch(0, nv_draw);
      __syncthreads();
      double best = 0; /* wave 0 */
      int flip = 0;
      for (int b0 = 0; b0 < RS_NVS_SAMPLES; b0 += bs, flip ^= 1) {
        const int nb = RS_NVS_SAMPLES - b0 < bs ? RS_NVS_SAMPLES - b0 : bs;
        const int total = nb * n;
        const uint8_t* cur = overlap ? nv_draw + flip * half : nv_draw;
        if (kDirect) {
          /* drop-in: the caller passes the rand() values it drew, in draw order */
          for (int j = tid; j < total; j += nt) nv_draw[j] = (uint8_t)(p.draws[(size_t)b0 * n + j] & 3);
          __syncthreads();
        } else if (!overlap) {
          if (wave == quota_wave) draw_batch(b0, nv_draw);
          __syncthreads();
        }
        if (overlap && wave == quota_wave) {
          if (b0 + bs < RS_NVS_SAMPLES) draw_batch(b0 + bs, nv_draw + (flip ^ 1) * half);
        } else {
          const int first = tid, step = overlap ? nt - 64 : nt; /* the generator is the last wave */
          for (int it = first; it < nb * R; it += step) {
            const int sl = idiv_small(it, R), r = it - sl * R;
            const uint8_t* row = s_cqi + r * Upad + ub;
            const uint8_t* dr = cur + sl * n;
            double hm = -1.0;
            int ha = 0xFFFF;
            int i = 0;
            for (; i + 4 <= n; i += 4) { /* AssignRBsGivenMCS :508-527, four users per step: the loads first */
              int d[4], h[4], cq[4];
              double vv[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) { d[q] = dr[i + q]; h[q] = nv_high[i + q]; cq[q] = row[i + q]; }
#pragma unroll
              for (int q = 0; q < 4; ++q) vv[q] = nv_val[(i + q) * 4 + d[q]];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int mc = h[q] - d[q] > 1 ? h[q] - d[q] : 1;
                const double metric = mc <= cq[q] ? vv[q] : 0.0;
                if (hm < metric) { hm = metric; ha = ub + i + q; }
              }
            }
            for (; i < n; ++i) {
              const int d = dr[i], h = nv_high[i];
              const int mc = h - d > 1 ? h - d : 1;
              const double metric = mc <= (int)row[i] ? nv_val[i * 4 + d] : 0.0;
              if (hm < metric) { hm = metric; ha = ub + i; }
            }
            nv_hm[it] = hm;
            nv_ha[it] = (uint16_t)ha;
          }
        }
        __syncthreads();
        if (wave == 0) {
          double pf = -1.0;
          if (lane < nb) {
            pf = 0;
            for (int r = 0; r < R; ++r) pf += nv_hm[lane * R + r];
          }
          /* scores are >= 0 (-1 on unused lanes): they order like (high word signed, low word unsigned) */
          const int hi = __double2hiint(pf);
          const int lo = (int)((unsigned)__double2loint(pf) ^ 0x80000000u);
          const int mhi = wave_max(hi);
          const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
          const double mx = __hiloint2double(mhi, (int)((unsigned)mlo ^ 0x80000000u));
          if (best < mx) { /* :442-446: the first sample that reaches the new maximum */
            best = mx;
            const int win = __ffsll((long long)__ballot(lane < nb && hi == mhi && lo == mlo)) - 1;
            if (lane < R) s_best_user[lane] = nv_ha[win * R + lane];
          }
        }
        __syncthreads();
      }
    }

    /* ---------------- P3: best user of every (RBG, segment) ---------------- */
    /* sched 7: the served slice in 8-aligned runs of nvs_seg users, so that a slice of a few dozen users keeps several waves
     * busy instead of R lanes; the run winners meet in P4 */
    const int nvs_seg = FIXED ? kCv.nvs_seg : p.nvs_seg;
    int nvs_lo = 0, nvs_hi = 0, nvs_first = 0;
    const bool nvs_split = SCHED == 7 && nvs_seg != 0;
    if (nvs_split) {
      nvs_lo = kDirect ? 0 : m->seg_begin[seg_lo];
      nvs_hi = kDirect ? U : m->seg_begin[seg_lo + 1];
      nvs_first = nvs_lo & ~7;
      nvs_runs = idiv_small(nvs_hi - nvs_first + nvs_seg - 1, nvs_seg);
    }
    const int n_items = nvs_split ? R * nvs_runs : o.n_items;
    /* one work item = (segment, RBG): winner to bu_out[it], its record (transport schedulers) to rec_out */
    auto scan_item = [&](int it, uint16_t* bu_out, uint32_t* rec_out, auto blk_tag) -> bool {
      bool held = false;
      float top1 = 0.0f, top2 = 0.0f; /* held winners: the two largest stage-1 values of the segment */
      {
        int sg = it / R, r = it - sg * R; /* r fastest: neighbouring lanes read neighbouring CQI bytes */
        int seg = SCHED == 7 ? seg_lo : sg;
        int ub = m->seg_begin[seg], ue = m->seg_begin[seg + 1];
        if (SCHED == 7 && kDirect) { ub = 0; ue = U; }
        if (nvs_split) {
          ub = nvs_first + sg * nvs_seg;
          ue = ub + nvs_seg;
          ub = ub < nvs_lo ? nvs_lo : ub;
          ue = ue > nvs_hi ? nvs_hi : ue;
        }
        double best = SCHED == 1 ? 0.0 : (SCHED == 7 ? -1.7976931348623157e308 : -1.0);
        int bu = -1;
        float run_a = 0.0f; /* stage-1 value of the leader (0: none yet) */
        bool exact = false; /* `best` is the leader's exact metric */
        int sl_eps = 1, sl_psi = 1;
        int sl_custom = 0; /* 1: alpha slice, 2: alpha slice with the HoL factor */
        int sl_id = 0;
        if (SCHED != 1) {
          int sl = SCHED == 7 ? (kDirect ? (int)p.user_slice[0] : seg) : seg;
          sl_id = sl;
          sl_eps = m->eps_psi[sl] & 1;
          sl_psi = (m->eps_psi[sl] >> 1) & 1;
          if (queue_mode_in && (m->eps_psi[sl] & 4)) sl_custom = (SCHED == 7 || (m->eps_psi[sl] & 8) != 0) ? 2 : 1;
        }
        const uint8_t* rowp = s_cqi + r * Upad;
        /* Exact two-stage argmax.  Stage 1 ranks the segment's users by the cheap
         * FP32 product a~ = fl32(num) * rcp32(fl32(den)) (v_rcp_f32, 1 ulp), which is within 2^-21.4 (relative) of the
         * reference's rounded FP64 quotient q = fl(num/den); a user whose a~ is below (1 - 2^-19) of
         * the largest a~ has a strictly smaller q and can neither win nor tie.  Stage 2 evaluates
         * the survivors with the real IEEE FP64 division, ascending user order, strict '>'.
         * Stage 1 reads 8 users per step (one 8-byte CQI load, two 16-byte reciprocal loads), keeps
         * the 32 products of a block in registers, takes their maximum, then marks the survivors. */
        const float kTol = 0x1.ffffcp-1f; /* 1 - 2^-19 */
        /* sched 7 scans runs of 8..32 users: a shape-specialised build ranks exactly one run per block */
        /* (the speculating schedulers carry this scan twice, here and in the serial phase: 32 products per block would spill) */
        constexpr int kP3Block = decltype(blk_tag)::v != 0 ? decltype(blk_tag)::v
                                 : (SCHED == 7 && FIXED && kCv.nvs_seg != 0) ? kCv.nvs_seg
                                 : ((kSpecSched && RS_P3_BLOCK > 16) ? 16 : RS_P3_BLOCK);
        const bool one_num = SCHED != 1 && !sl_eps;
        const float* numtab = one_num ? m->ones16 : s_num32; /* a table either way: no branch per user */
        /* this slice's window, indexed by user (drop-in NVS passes the served slice's users only: their own slice id) */
        const float* rcw = s_rcp32 + (SCHED == 1 ? 0 : m->rcp_off[(SCHED == 7 && kDirect) ? (int)p.user_slice[0] : seg]);
        /* stage 2 of one user: the reference's expression, real IEEE FP64 division */
        auto exact_metric = [&](int u, int c) -> double {
          if (SCHED == 1) {
            /* ref: dl-pf-packet-scheduler.cpp:128-140  (se*180000.)/avg */
            return s_num[c] / s_avg[u];
          }
          /* ref: :688-693  pow(se_kbps, eps) / pow(avg_kbps, psi), eps, psi in {0,1} */
          /* both table reads are issued whatever the slice's exponents are (no branch around an LDS read) */
          const double num_c = s_num[c], den_u = s_avgk[u];
          double num = sl_eps ? num_c : 1.0, den = sl_psi ? den_u : 1.0;
          if constexpr (DIRECT) {
            /* pow(se_kbps, epsilon) / pow(avg_kbps, psi) for any integers (ref: :690-693): both powers come from the host's libm
             * -- 16 numerators per slice at rs_create, the denominator of every user per call -- the device only divides */
            if (p.gen_exp) { num = p.gen_num[sl_id * 16 + c]; den = den_u; }
          }
          if (QUEUE && (prio_in[u] & 2) == 0) return -2.0; /* not in UsersToSchedule: below the scan's start value of -1 */
          /* (schedulers 1 and 7 with queues do not come here: serial allocator below) */
          if (sl_custom && prio_in && (prio_in[u] & 1) == 0) return 0.0;
          if (sl_custom == 2) return hol_in[u] * num / den; /* HoL * pow(se) / pow(avg), left to right */
          return num / den;
        };
        if constexpr (DIRECT) {
          /* The caller's averages / head-of-line delays are arbitrary doubles (the batch EWMA keeps its own in [1, ~1e12]): when
           * one of them is not an ordinary FP32 number -- huge, tiny, negative, infinite, NaN -- the stage-1 error bound does not
           * hold, the host says so (rs_schedule_tti), and every user of the segment is compared with the reference's expression
           * itself, ascending, strict '>' (a NaN metric never wins, as in the reference's scan). */
          if (p.exact_scan) {
            for (int u = ub; u < ue; ++u) {
              const double metric = exact_metric(u, rowp[u]);
              if (metric > best) { best = metric; bu = u; }
            }
            exact = true;
            ub = ue; /* nothing left for the block loop */
          }
        }
        /* (an empty segment must not enter: its 8-aligned start lies before its end, inside a neighbour's window) */
        for (int blk = ue > ub ? (ub & ~7) : ue; blk < ue; blk += kP3Block) {
          /* a~ of the 32 users blk..blk+31 (0 for users outside [ub, ue) and for the padding) */
          float av[kP3Block];
          float best_a = 0.0f;
#pragma unroll
          for (int g = 0; g < kP3Block / 8; ++g) {
            const int u0 = blk + 8 * g;
            uint2 cw = make_uint2(0u, 0u);
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (u0 < ue) {
              cw = *(const uint2*)(rowp + u0);
              ra = *(const float4*)(rcw + u0);
              rb = *(const float4*)(rcw + u0 + 4);
            }
            const float rc[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            float nm[8];
            /* CQI bytes are <= 15, so four of them scale to table byte offsets with one shift and each address is
             * one byte-select add */
            const uint32_t cx4 = cw.x << 2, cy4 = cw.y << 2;
#pragma unroll
            for (int k = 0; k < 8; ++k)
              nm[k] = *(const float*)((const char*)numtab + (((k < 4 ? cx4 : cy4) >> (8 * (k & 3))) & 0xffu));
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(nm[k])); /* eight table reads in flight, none behind a branch */
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              /* no range test: a slot outside the slice (or, sched 1, beyond the last user) reads a zero reciprocal */
              const float a = nm[k] * rc[k];
              av[8 * g + k] = a;
              best_a = fmaxf(best_a, a);
              if constexpr (kHoldSched) {
                top2 = fmaxf(top2, fminf(top1, a));
                top1 = fmaxf(top1, a);
              }
            }
          }
          if (!sl_custom) {
            /* The leader so far is known by its stage-1 value run_a; its exact metric is only worked out when somebody comes
             * within the tolerance of it.  A block whose maximum is below (1 - 2^-19) run_a cannot win or tie; otherwise the
             * users at or above (1 - 2^-19) of the larger of the two maxima are the only possible winners: a single one that
             * leaves the old leader below the threshold simply takes over (no division at all -- the common case), several, or
             * one next to a leader still in range, are compared exactly, ascending user order, strict '>'. */
            if (best_a > 0.0f && best_a >= run_a * kTol) {
              const float thr = fmaxf(best_a, run_a) * kTol;
              uint32_t cand = 0;
#pragma unroll
              for (int k = 0; k < kP3Block; ++k) cand |= av[k] >= thr ? (1u << k) : 0u;
              const bool leader_in = run_a >= thr; /* false while there is no leader: thr > 0 */
              if (!leader_in && (cand & (cand - 1u)) == 0u) {
                bu = blk + __ffs((int)cand) - 1;
                run_a = best_a;
                exact = false;
              } else {
                if (leader_in) {
                  if (!exact) best = exact_metric(bu, rowp[bu]);
                } else {
                  best = SCHED == 1 ? 0.0 : (SCHED == 7 ? -1.7976931348623157e308 : -1.0);
                  bu = -1;
                }
                while (cand) {
                  const int j = __ffs((int)cand) - 1;
                  cand &= cand - 1;
                  const int u = blk + j;
                  const double metric = exact_metric(u, rowp[u]);
                  if (metric > best) { best = metric; bu = u; }
                }
                exact = true;
                run_a = *(const float*)((const char*)numtab + ((uint32_t)rowp[bu] << 2)) * rcw[bu];
              }
            }
          } else {
            /* customised slices (drop-in mode only): every survivor of the block is evaluated.  Survivors: a~ >= (1 - 2^-19)
             * max a~ and a~ > 0; with an all-zero block an infinite threshold leaves no survivor */
            const float thr = best_a > 0.0f ? best_a * kTol : __builtin_inff();
            uint32_t cand = 0;
#pragma unroll
            for (int k = 0; k < kP3Block; ++k) cand |= av[k] >= thr ? (1u << k) : 0u;
            /* customised slices can rank every user at 0 (no prioritized data): the reference's scan then
             * keeps the first user (0 > -1), so that user goes to stage 2 */
            if (!cand) {
              if (QUEUE) { /* users without queued data are not in the list: every slot of the slice, the first listed one wins */
#pragma unroll
                for (int k = 0; k < kP3Block; ++k) cand |= (blk + k >= ub && blk + k < ue) ? (1u << k) : 0u;
              } else {
                cand = 1u << ((ub > blk ? ub : blk) - blk);
              }
            }
            while (cand) {
              const int j = __ffs((int)cand) - 1;
              cand &= cand - 1;
              const int u = blk + j;
              const double metric = exact_metric(u, rowp[u]);
              if (metric > best) { best = metric; bu = u; }
            }
            exact = true;
          }
        }
        const int bkey = bu >= 0 ? rowp[bu] : 0;
        if ((SCHED == 1 || nvs_split) && !exact && bu >= 0) best = exact_metric(bu, bkey); /* the winners' metrics meet in P4 */
        bu_out[it] = (uint16_t)bu;
        if constexpr (kHoldSched) {
          /* Held until the next full scan?  The winner's stage-1 value must BE the segment's largest and lead the second
           * largest by mu = 2^-18 + 2 / (1 + avg_w) (2^-18 covers the stage-1 error on both sides, 2 / (1 + avg_w)
           * what RS_HOLD_MAX_AGE unserved TTIs take from the winner through the "+1" of (1 + avg) / 1000; avg_w >= 64 keeps the
           * clamp at 1 out of reach).  A psi = 0 slice ranks on the CQI alone: its winner stands whoever is served. */
          if (bu >= 0) {
            if (!sl_psi) {
              held = true;
            } else {
              const float la = *(const float*)((const char*)numtab + ((uint32_t)bkey << 2)) * rcw[bu];
              const double aw = s_avg[bu];
              const float mu = 0x1p-18f + 2.0f / (1.0f + (float)aw);
              held = aw >= 64.0 && la >= top1 && top2 * (1.0f + mu) * 1.000001f <= la;
            }
          }
        }
        if (SCHED == 10) {
          /* UpperBound sorts one vector per slice (:229-233): slice-major */
          rec_out[sg * R + r] = ((uint32_t)bkey << 16) | ((uint32_t)r << 8) | (uint32_t)sg;
        } else if (kTransport) {
          /* MaximizeCell's vector is RBG-major, slice-minor (:357-360) */
          rec_out[r * S + sg] = ((uint32_t)bkey << 16) | ((uint32_t)r << 8) | (uint32_t)sg;
        } else if (SCHED == 1 || nvs_split) {
          s_best_metric[it] = best;
        }
      }
      return held;
    };
    if constexpr (kQSerial && SCHED == 7) {
      /* m_requiredRBs (packet-scheduler.cpp:319-334): the data of the bearer that created the user's record, in PRBs of the
       * wideband MCS -- EESM over every PRB of the band in order, G identical terms per RBG */
      const int ub = m->seg_begin[seg_lo], ue = m->seg_begin[seg_lo + 1];
      for (int u = ub + tid; u < ue; u += nt) {
        int need = 0;
        if (prio_in[u] & 2) {
          double sum = 0;
          if (per_prb) { /* per-PRB reports: every PRB of the band as the user reported it */
            for (int r = 0; r < R; ++r) {
              const uint8_t* pr = prb_ptr(u, r);
              for (int k = 0; k < G; ++k) sum += s_e[pr[k]];
            }
          } else {
            for (int r = 0; r < R; ++r) {
              const double ev = s_e[s_cqi[r * Upad + u]];
              for (int k = 0; k < G; ++k) sum += ev;
            }
          }
          const double x = sum / (double)(R * G);
          int wide = 15;
          if (!(x == 0)) {
            wide = 1;
#pragma unroll
            for (int k = 1; k <= 13; ++k) wide += (x <= xthr_k[k - 1]) ? 1 : 0;
          }
          const int first = q_data0[u] > 0 ? q_data0[u] : q_data1[u];
          need = (first * 8) / tab->tbs1_of_cqi[wide];
        }
        q_need[u] = need;
      }
      /* The metric of a (user, RBG) pair depends on the RBG through the CQI only: 16 quotients per user of the served slice,
       * divided here by all threads, instead of one division per candidate and RBG on the serial wave (the idle sort buffers
       * hold the table when the slice fits: 128 bytes per user) */
      if (ue - ub <= 64 && (ue - ub) * 128 <= o.items - o.elems) {
        const int sl_eps7 = m->eps_psi[seg_lo] & 1, sl_psi7 = (m->eps_psi[seg_lo] >> 1) & 1;
        const bool custom7 = (m->eps_psi[seg_lo] & 4) != 0;
        double* const qt = (double*)s_elems;
        for (int i = tid; i < (ue - ub) * 16; i += nt) {
          const int u = ub + (i >> 4), cq = i & 15;
          const double num = sl_eps7 ? s_num[cq] : 1.0, den = sl_psi7 ? s_avgk[u] : 1.0;
          qt[i] = !custom7 ? num / den : ((prio_in[u] & 1) == 0 ? 0.0 : hol_in[u] * num / den); /* ref: nvs :375-387 */
        }
      }
    }
    bool hold_full = true; /* held winners: this TTI scanned every item (their records are in cur_rec) */
    if constexpr (kHoldSched) {
      /* ---- held winners: scan only what can have changed ---- */
      /* The items to scan again -- winner not held, or served in the previous TTI (a psi = 0 slice ignores the averages) -- are
       * packed into a wave-private list (items are dealt to the waves round robin: no shared list, no atomics, no workgroup
       * barrier) and scanned FOUR LANES PER ITEM,
       * 8 (or 16) users of the slice's zero-padded window per lane: the stage-1 products as in scan_item, the window's two
       * largest by two-step butterflies inside the lane quad, then the lane that holds the only user within 2^-19 of the
       * largest writes the winner (no division); several such users are compared exactly, ascending, strict '>'.
       * ~60 of 500 items per TTI at the benchmark's shape: one pass of 16 items on every wave. */
      bool full = !hold_ok || tti == 0 || grid_loaded || hold_age >= RS_HOLD_MAX_AGE;
      int hold_listed = 0;
#if defined(RS_STAMPS) && defined(RS_STAMPS_HOLD)
      unsigned long long hs_prev = __builtin_readcyclecounter();
#define RS_HSTAMP(i) do { if (tid == 0) { unsigned long long n_ = __builtin_readcyclecounter(); sort_sub[i] += n_ - hs_prev; hs_prev = n_; } } while (0)
#else
#define RS_HSTAMP(i) do { } while (0)
#endif
      if (!full) {
        const float kTolH = 0x1.ffffcp-1f; /* 1 - 2^-19 */
        const int q4 = lane & 3, grp = lane >> 2;
        /* groups of 8 users per lane: one for windows of up to 32 users, two up to 64 (a compile-time fact when the host passed
         * the batch's longest window) */
        constexpr int kMaxGrp = (FIXED && RS_JIT_WIN > 0 && RS_JIT_WIN <= 32) ? 1 : 2;
        const int ngrp = kMaxGrp == 1 ? 1 : (hold_win <= 32 ? 1 : 2);
        auto quad_max_i = [&](int v) -> int { /* every lane of the quad gets the quad's maximum: quad_perm [1,0,3,2], [2,3,0,1] */
          int o1 = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
          v = v > o1 ? v : o1;
          int o2 = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
          return v > o2 ? v : o2;
        };
        uint16_t* const wl = hold_wl; /* this wave's list */
        for (int k0 = 0; k0 * nwaves < n_items; k0 += 64) {
          /* (a one-chunk shape: the list may have been packed in the previous TTI's serial phase already) */
          const int n_list = (pre_listed >= 0 && k0 == 0) ? pre_listed : hold_pack(k0, wave);
          hold_listed += n_list;
          RS_HSTAMP(0);
          /* issue priority of the pass (same-box A/B, 512 cells): GreedyByRow 101.4 against 99.7 M TTIs/s at priority 1 -- the
           * co-resident cell is mostly in its one-wave serial phase, which runs at 3 anyway; MaximizeCell 32.8 against 33.7 -- there
           * the pass would take issue slots from the co-resident cell's sort levels (priority 1, a barrier every few dozen
           * instructions) */
          constexpr int kHoldPrio = SCHED == 8 ? 1 : 0;
          if (kHoldPrio) __builtin_amdgcn_s_setprio(kHoldPrio);
          for (int base = 0; base < n_list; base += 16) {
            const bool on = base + grp < n_list;
            const int it = on ? (int)wl[base + grp] : 0;
            const int sg = FIXED ? it / RS_JIT_R : idiv_small(it, R), r = it - sg * R;
            const int ub = m->seg_begin[sg], ue = m->seg_begin[sg + 1];
            const int bits = m->eps_psi[sg];
            const float* numtab = (bits & 1) ? s_num32 : m->ones16;
            const float* rcw = s_rcp32 + m->rcp_off[sg];
            const uint8_t* rowp = s_cqi + r * Upad;
            const int wend = (ue + 7) & ~7;
            const int u0 = (ub & ~7) + q4 * 8 * ngrp;
            float t1 = 0.0f, t2 = 0.0f; /* my users' largest and second largest stage-1 value */
            float av[8 * kMaxGrp];
#pragma unroll
            for (int g = 0; g < kMaxGrp; ++g) {
              const int ug = u0 + 8 * g;
              uint2 cw = make_uint2(0u, 0u);
              float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
              if (on && g < ngrp && ug < wend) {
                cw = *(const uint2*)(rowp + ug);
                ra = *(const float4*)(rcw + ug);
                rb = *(const float4*)(rcw + ug + 4);
              }
              const float rc[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
              const uint32_t cx4 = cw.x << 2, cy4 = cw.y << 2;
              float nm[8];
#pragma unroll
              for (int k = 0; k < 8; ++k)
                nm[k] = *(const float*)((const char*)numtab + (((k < 4 ? cx4 : cy4) >> (8 * (k & 3))) & 0xffu));
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const float a = nm[k] * rc[k]; /* 0 outside the slice: zero reciprocals around it */
                av[8 * g + k] = a;
                t2 = fmaxf(t2, fminf(t1, a));
                t1 = fmaxf(t1, a);
              }
            }
            /* the quad's two largest: (t1, t2) pairs meet in two butterfly steps */
            float g1 = t1, g2 = t2;
#pragma unroll
            for (int step = 0; step < 2; ++step) {
              const float o1 = __int_as_float(step == 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(g1), 0xB1, 0xf, 0xf, false)
                                                        : __builtin_amdgcn_update_dpp(0, __float_as_int(g1), 0x4E, 0xf, 0xf, false));
              const float o2 = __int_as_float(step == 0 ? __builtin_amdgcn_update_dpp(0, __float_as_int(g2), 0xB1, 0xf, 0xf, false)
                                                        : __builtin_amdgcn_update_dpp(0, __float_as_int(g2), 0x4E, 0xf, 0xf, false));
              g2 = fmaxf(fmaxf(g2, o2), fminf(g1, o1));
              g1 = fmaxf(g1, o1);
            }
            /* my users within the tolerance of the window's largest */
            /* (an all-zero window -- only an empty slice, which is never listed -- gets an unreachable threshold instead of a
             * second comparison per user) */
            const float thr = g1 > 0.0f ? g1 * kTolH : __builtin_inff();
            unsigned cm = 0u;
#pragma unroll
            for (int k = 0; k < 8 * kMaxGrp; ++k) cm |= av[k] >= thr ? (1u << k) : 0u;
            int cnt = __popc(cm);
            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, false);
            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, false);
            bool writer = on && cnt == 1 && cm != 0u;
            int ul = u0 + __ffs((int)cm) - 1; /* (writer: my only survivor) */
            float la = t1;
            if (__ballot(on && cnt > 1) != 0ull) {
              /* several users within the tolerance of each other: the reference's expression, first maximum */
              int bhi = -1, blo = (int)0x80000000, bu_l = 0;
              if (on && cnt > 1) {
                unsigned c2 = cm;
                while (c2) {
                  const int j = __ffs((int)c2) - 1;
                  c2 &= c2 - 1u;
                  const int u = u0 + j;
                  const double metric = ((bits & 1) ? s_num[rowp[u]] : 1.0) / ((bits & 2) ? s_avgk[u] : 1.0);
                  const int hi = __double2hiint(metric), lo = (int)((unsigned)__double2loint(metric) ^ 0x80000000u);
                  if (hi > bhi || (hi == bhi && lo > blo)) { bhi = hi; blo = lo; bu_l = u; } /* metrics > 0: (high, low unsigned) order */
                }
              }
              const int mhi = quad_max_i(bhi);
              const int mlo = quad_max_i(bhi == mhi ? blo : (int)0x80000000);
              const unsigned long long eq = __ballot(on && cnt > 1 && bhi == mhi && blo == mlo);
              const unsigned quad = (unsigned)(eq >> (lane & ~3)) & 0xfu; /* lanes of my quad that reach the maximum: the lowest wins */
              if (on && cnt > 1 && (quad & ((1u << q4) - 1u)) == 0u && ((quad >> q4) & 1u)) {
                writer = true;
                ul = bu_l;
                la = numtab[rowp[bu_l]] * rcw[bu_l];
              }
            }
            if (writer) {
              const int key = rowp[ul];
              bool held = false;
              if (!(bits & 2)) {
                held = true;
              } else {
                const double aw = s_avg[ul];
                const float mu = 0x1p-18f + 2.0f / (1.0f + (float)aw);
                held = aw >= 64.0 && la >= g1 && g2 * (1.0f + mu) * 1.000001f <= la;
              }
              cur_bu[it] = (uint16_t)ul;
              /* (MaximizeCell rebuilds its records from the winners right before the sort: the sort permutes them in place) */
              if (SCHED != 9) cur_rec[r * S + sg] = ((uint32_t)key << 16) | ((uint32_t)r << 8) | (uint32_t)sg;
              /* (item = k * nwaves + wave: bit k & 63 of this wave's word k >> 6; the quads of one wave write it side by side) */
              const int kk = FIXED ? it / (RS_JIT_NT / 64) : idiv_small(it, nwaves);
              unsigned long long* const word = &hold_bits[(kk >> 6) * nwaves + wave];
              if (held) atomicOr(word, 1ull << (kk & 63));
              else atomicAnd(word, ~(1ull << (kk & 63)));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          if (kHoldPrio) __builtin_amdgcn_s_setprio(0);
          RS_HSTAMP(1);
        }
        hold_age += 1;
      }
      if (full) {
        for (int k0 = 0; k0 * nwaves < n_items; k0 += 64) { /* the same dealing of the items as above */
          const int j = (k0 + lane) * nwaves + wave;
          bool held = false;
          if (j < n_items) held = scan_item(j, cur_bu, cur_rec, RsInt<RS_P3_BLOCK_TOP>{});
          const unsigned long long hm = __ballot(held);
          if (hold_ok && lane == 0) hold_bits[(k0 >> 6) * nwaves + wave] = hm;
        }
        hold_age = 0;
      }
      hold_full = full;
      pre_listed = -1;
#ifdef RS_STAMPS
      if (tid == 0) sort_sub[6] += (unsigned long long)hold_listed + ((unsigned long long)(full ? 1 : 0) << 32); /* (wave 0's chunk) */
#endif
    } else
    if constexpr (SCHED != 11 && !kQSerial) {
     if (!(kEarly17 && have_scan)) { /* (NVS: the winners may be in place already, found during the previous TTI's serial phase) */
      /* fix-up of a speculated TTI: only the items whose speculative winner was served in the previous TTI (listed by the
       * scanning waves) are scanned again, now with the true averages; a list that overflowed means all of them.  One loop
       * for both cases: the scan is inlined once here and once in the serial phase. */
      /* The serial phase scans whole rounds only (every scanning lane one item per round): the items beyond them were not
       * speculated at all and are scanned here, behind the listed ones -- a second, nearly empty round in the serial phase
       * would cost a whole item's latency there, here they ride along in lanes that are idle anyway. */
      const int n_fix = have_spec ? rs_lds_load(&fl_prev->n_fix) : 0;
      const bool listed = have_spec && n_fix <= RS_FIX_CAP;
      const int n_spec = spec_items(n_items); /* items the serial phase speculated: [0, n_spec) */
      const int n_scan = listed ? n_fix + (n_items - n_spec) : n_items;
      for (int j = tid; j < n_scan; j += nt)
        scan_item(listed ? (j < n_fix ? (int)fix_list[j] : n_spec + (j - n_fix)) : j, cur_bu, cur_rec, RsInt<RS_P3_BLOCK_TOP>{});
     }
    }
    /* (an NVS TTI whose averages, slice and winners were all prepared has written nothing since the previous TTI's closing barrier) */
    if (!(kEarly17 && SCHED == 7 && have_scan)) __syncthreads();
    if (kTransport && p.log_keys) {
      /* parity tests only: what the inter-slice step is about to read, [R][S] per TTI: CQI key of the slice's best user
       * (0: no user) | (user + 1) << 8 -- flow_spectraleff / user_index of ref :545-567 */
      for (int i = tid; i < R * S; i += nt) {
        const int r = SCHED == 10 ? i % R : i / S, sg = SCHED == 10 ? i / R : i % S;
        const int bu = cur_bu[sg * R + r];
        const uint32_t key = kHoldSched ? (bu == 0xFFFF ? 0u : (uint32_t)s_cqi[r * Upad + bu]) : (cur_rec[i] >> 16);
        p.log_keys[((size_t)cell * p.n_ttis + tti) * R * S + r * S + sg] = key | ((bu == 0xFFFF ? 0u : (uint32_t)bu + 1u) << 8);
      }
    }
    RS_STAMP(2);

    /* ---------------- P4: inter-slice assignment ---------------- */
    if (SCHED == 9) {
      const int N = R * S;
      /* std::sort emulation (:361): introsort loop, then the final insertion sort */
      {
        uint16_t* sx = (uint16_t*)(lds + o.sortx);
        uint16_t* pa = (uint16_t*)s_sorted;
        /* EPT = array positions per thread, picked by the host (0: any size, state in LDS) */
        /* Issue priority rises towards the serial end of the TTI: 0 for the throughput phases (EWMA, metric scan), 1 for the
         * barrier-paced sort levels, 2 for the counting sort, 3 for the one wave that runs the greedy scan and the link
         * adaptation.  A phase that meets a barrier every few dozen instructions loses most when the co-resident cell's
         * waves interleave with it (measured with two cells per CU: +5..6 % from the sort's priority alone). */
        __builtin_amdgcn_s_setprio(1);
        if constexpr (kHoldSched) {
          if (!hold_full) {
            /* a TTI that scanned only the listed items: every thread rebuilds the records of the positions it owns in the sort
             * (the previous sort permuted the array in place) from the winners and the CQI grid */
            for (int x = tid; x < N; x += nt) {
              const int r = FIXED ? x / RS_JIT_S : idiv_small(x, S), sg = x - r * S;
              const int bu = cur_bu[sg * R + r];
              const uint32_t key = bu == 0xFFFF ? 0u : (uint32_t)s_cqi[r * Upad + bu];
              s_elems[x] = (key << 16) | ((uint32_t)r << 8) | (uint32_t)sg;
            }
            if (EPT == 0) __syncthreads(); /* (the LDS form of the sort reads other positions first) */
          }
        }
        if constexpr (EPT > 0) introsort_levels_reg<EPT>(s_elems, N, s_sorted, (int32_t*)sx, m, sort_sub);
        else introsort_loop_levels(s_elems, N, pa, pa + N, sx, sx + N, sx + 2 * N, sx + 3 * N, m);
      }
      RS_STAMP(3);
      __builtin_amdgcn_s_setprio(2);
      if constexpr (EPT > 0) counting_sort_desc_owned<EPT>(s_elems, s_sorted, N, m);
      else counting_sort_desc(s_elems, s_sorted, N, m);
      __builtin_amdgcn_s_setprio(0);
      RS_STAMP(4);
    }
    if constexpr (SCHED == 10) {
      /* ---------------- UpperBound, ref: :223-246 and the inter_sched_ >= 4 branch of :603-616 ----------------
       * Every slice with a positive quota sorts its own R (rbg, eff) pairs (the same unstable std::sort) and takes its
       * first quota RBGs whatever the other slices take.  The S sorts run as ONE level-synchronous pass over the
       * slice-major array (sub-ranges of a level are disjoint anyway); the final insertion sort of each call is a stable
       * rank inside its segment.  Then one thread per taken (slice, k) entry: the first entry of a UE is its leader and
       * sums E[cqi] over the UE's entries in push order (the slice's sorted order, not RBG order). */
      static_assert(EPT > 0, "UpperBound uses the register form of the sort (R*S <= 4 * threads)");
      const int N = R * S;
      __builtin_amdgcn_s_setprio(1);
      introsort_levels_reg<EPT>(s_elems, N, s_sorted, (int32_t*)(lds + o.sortx), m, sort_sub, R);
      __builtin_amdgcn_s_setprio(0);
      int32_t* low_owner = (int32_t*)m->hist; /* per RBG: (slice << 16 | UE) of the lowest slice holding it */
      for (int x = tid; x < N; x += nt) {
        const int f = idiv_small(x, R) * R;
        const uint32_t e = s_elems[x];
        const int k = (int)(e >> 16);
        int rank = 0;
        for (int y = f; y < f + R; ++y) {
          const int ky = (int)(s_elems[y] >> 16);
          rank += (ky > k || (ky == k && y < x)) ? 1 : 0;
        }
        s_sorted[f + rank] = e;
      }
      if (tid < R) low_owner[tid] = 0x7fffffff;
      if (tid == 0) m->served = 0;
      __syncthreads();
      uint16_t* ent_user = (uint16_t*)s_elems; /* s_elems is dead: UE of entry (slice, k), 0xFFFF = not taken */
      for (int x = tid; x < N; x += nt) {
        const int sl = idiv_small(x, R), k = x - sl * R;
        const int q = m->quota[sl];
        int u = 0xFFFF;
        if (k < q) { /* q <= 0: nothing; q > R cannot index past the segment (the reference would read past its vector) */
          const int rbg = (int)((s_sorted[x] >> 8) & 63u);
          u = s_best_user[sl * R + rbg];
          if (u != 0xFFFF) atomicMin(&low_owner[rbg], (sl << 16) | u);
        }
        ent_user[x] = (uint16_t)u;
        if (p.log_upper) p.log_upper[x] = u == 0xFFFF ? -1 : (int)((s_sorted[x] >> 8) & 63u) | (u << 8);
      }
      __syncthreads();
      for (int x = tid; x < N; x += nt) {
        const int sl = idiv_small(x, R), f = sl * R;
        const int u = ent_user[x];
        int q = m->quota[sl];
        q = q > R ? R : q;
        bool leader = u != 0xFFFF;
        for (int y = f; y < x && leader; ++y) leader = ent_user[y] != u;
        if (!leader) continue;
        double sum = 0;
        int nprb = 0, syn_bits = 0; /* syn_bits: the synthetic-experiment transport block, every PRB at its own CQI (:653-659) */
        for (int y = x; y < f + q; ++y) {
          if (ent_user[y] != u) continue;
          const int r2 = (int)((s_sorted[y] >> 8) & 63u);
          if (per_prb) {
            const uint8_t* pr = prb_ptr(u, r2);
            for (int g = 0; g < G; ++g) { sum += s_e[pr[g]]; syn_bits += m->tbs1_of_cqi[pr[g]]; }
          } else {
            const int cq = s_cqi[r2 * Upad + u];
            const double ev = s_e[cq];
            for (int g = 0; g < G; ++g) sum += ev;
            syn_bits += G * m->tbs1_of_cqi[cq];
          }
          nprb += G;
        }
        const double xm = sum / (double)nprb;
        int fcqi;
        if (xm == 0) {
          fcqi = 15;
        } else {
          fcqi = 1;
          for (int t = 1; t <= 13; ++t) fcqi += (xm <= s_x[t]) ? 1 : 0;
        }
        const int mcs = m->mcs_of_cqi[fcqi];
        const int tbs = p.synthetic ? syn_bits : s_tbs[(nprb / G) * 16 + fcqi];
        int bytes = tbs / 8;
        if (bytes > 100000000) bytes = 100000000;
        if (bytes > 0) {
          if (kCumRegs) {
            s_tx[u] += bytes | (nprb << RS_TX_NPRB_SHIFT);
          } else {
            s_tx[u] += bytes;
            atomicAdd((unsigned long long*)&p.cum_bytes[(size_t)cell * U + u], (unsigned long long)bytes);
            atomicAdd((unsigned long long*)&p.cum_rbs[(size_t)cell * U + u], (unsigned long long)nprb);
          }
        }
        atomicAdd(&m->served, 1);
        if (p.log_map) {
          const size_t row = (size_t)cell * p.n_ttis + tti;
          if (p.log_tbs) p.log_tbs[row * U + u] = tbs;
          if (p.log_uinfo) p.log_uinfo[row * U + u] = nprb | (fcqi << 16) | (mcs << 24);
        }
      }
      if (tid < S) {
        /* ref: :618-620 with slice_final_rbgs = size of the slice's list */
        int q = m->quota[tid];
        q = q < 0 ? 0 : (q > R ? R : q);
        s_sstate[tid] = (double)(m->target[tid] - q * G);
      }
      if (p.log_map) {
        const size_t row = (size_t)cell * p.n_ttis + tti;
        if (tid < R) p.log_map[row * R + tid] = (int16_t)(low_owner[tid] == 0x7fffffff ? -1 : (low_owner[tid] & 0xffff));
        if (tid < S) {
          if (p.log_quota) p.log_quota[row * S + tid] = (int16_t)m->quota[tid];
          if (p.log_target) p.log_target[row * S + tid] = (int16_t)m->target[tid];
        }
      }
    }
    /* the rest of the TTI runs on wave 0: lanes = slices for the quota counters, lanes = RBGs for
     * the allocation; the RBG->slice map stays in registers */
    /* Does the NEXT TTI refresh the CQI grid?  Then nothing can be prepared for it now (its grid is not in LDS yet). */
    bool spec_next = false;
    if (kSpecSched && spec_enabled && tti + 1 < p.n_ttis) {
      if (p.cqi_mode == RS_CQI_EPOCHS) {
        spec_next = epoch_pos + 1 != p.refresh;
      } else if (p.cqi_mode == RS_CQI_TRACE) {
        const double t_next = t + 0.001;
        spec_next = reported && !(((int)(t_next * 1000) - last_sent) >= 40);
      }
    }
    /* Held winners: the next TTI's quotas need nothing but this TTI's slice offsets (and, with the error model's draws on the
     * stream, the number of users served): the quota wave, idle during the serial phase, works them out there instead of at the
     * top of the next TTI, where every wave now has only a few items to scan and the quota wave would be the last to arrive */
    const bool quota_next = kHoldSched && nwaves >= 2 && tti + 1 < p.n_ttis;
    /* ... and so are the other waves: they apply the next TTI's EWMA decay to every user ((1 - beta) * avg exactly, as if nobody
     * were served; wave 0 adds beta * rate for the users it served once its link adaptation knows their bytes -- the reference's
     * sum of two rounded products, as in round 2's speculation) and, once the served set is published, pack their lists of items
     * to scan again.  The top of the next TTI is then one pass over those lists.
     * Same-box A/B (512 cells, 25 RBGs): GreedyByRow 94.8 against 93.4 M TTIs/s (its TTI is short: the top of the TTI is a large
     * share), MaximizeCell 32.86 against 33.16 -- the waves that work beside wave 0 slow its greedy scan and link adaptation by
     * what the shorter top saves, as round 2 found for its speculation -- so: GreedyByRow only. */
    const bool ewma_next = quota_next && SCHED == 8;
    /* schedulers 1 / 7: the other waves prepare TTI t+1 beside wave 0 (see kEarly17); not past the end of the launch */
    const bool early17 = kEarly17 && nwaves >= 2 && tti + 1 < p.n_ttis;
    bool early_scan_ok = false; /* NVS: TTI t+1 reads the CQI grid that is in LDS now */
    if (kEarly17 && SCHED == 7 && early17) {
      if (p.cqi_mode == RS_CQI_EPOCHS) {
        early_scan_ok = epoch_pos + 1 != p.refresh;
      } else if (p.cqi_mode == RS_CQI_TRACE) {
        const double t_next = t + 0.001;
        early_scan_ok = reported && !(((int)(t_next * 1000) - last_sent) >= 40);
      }
    }
    if (SCHED != 10 && wave == 0) {
      /* the only running wave of this cell until the end-of-TTI barrier: ask the SIMD's arbiter to prefer it
       * over the co-resident cell's waves (measured +3 % with two cells per CU) */
      __builtin_amdgcn_s_setprio(RS_SERIAL_PRIO);
      int owner = -1;
      int got = 0; /* lane s: RBGs granted to slice s */
      int my_target = 0, my_quota = 0; /* lane s: this TTI's values (the quota wave may overwrite the LDS copies for TTI t+1) */
      if (kHoldSched) {
        hold_served[lane] = 0u; /* (m->hist is free: this TTI's lists are consumed, the counting sort is over) */
      }
      if (kEarly17) {
        if (lane == 0) { fl_prev->ctr_p1 = 0; fl_prev->ctr_p3 = 0; fl_prev->greedy_done = 0; fl_prev->n_fix = 0; }
      }
      if (kSpecSched || kHoldSched) {
        my_target = m->target[lane];
        my_quota = m->quota[lane];
        /* the other flag set belongs to the NEXT serial phase: its last reader (this TTI's fix-up) is behind a barrier */
        if (lane == 0) { fl_prev->ctr_p1 = 0; fl_prev->ctr_p3 = 0; fl_prev->greedy_done = 0; fl_prev->n_fix = 0; }
        if (spec_next && lane < 32) served_bits[lane] = 0u;
      }
      if constexpr (kQSerial) {
        /* Schedulers 1 and 7 with finite queues: RBG by RBG, the first maximum among the candidates still in the race --
         * flows whose transport block does not yet carry their queue (downlink-packet-scheduler.cpp:221-265) / users of the
         * served slice below their m_requiredRBs (downlink-nvs-scheduler.cpp:283-308).  Lanes = candidates, 64 at a time;
         * metrics are >= 0, so their bit patterns order like (high word signed, low word unsigned): two DPP max reductions
         * and the lowest lane among the equals; a later chunk only wins with a strictly larger metric. */
        int c_lo = 0, c_hi = 2 * U; /* sched 1: flow ids 2 * user + bearer */
        int sl_eps = 1, sl_psi = 1, sl_custom = 0;
        if (SCHED == 7) {
          c_lo = m->seg_begin[seg_lo];
          c_hi = m->seg_begin[seg_lo + 1];
          sl_eps = m->eps_psi[seg_lo] & 1;
          sl_psi = (m->eps_psi[seg_lo] >> 1) & 1;
          sl_custom = (m->eps_psi[seg_lo] & 4) != 0;
        }
        /* sched 7, served slice of at most 64 users with its metric table in place: lane = user, the user's remaining
         * m_requiredRBs in a register, per RBG two LDS reads (the next RBG's CQI is fetched a step ahead) and the two reductions */
        const bool table7 = SCHED == 7 && c_hi - c_lo <= 64 && (c_hi - c_lo) * 128 <= o.items - o.elems;
        if (table7) {
          const int u = c_lo + lane;
          const bool mine = u < c_hi && (prio_in[u < c_hi ? u : c_lo] & 2) != 0;
          int need = mine ? q_need[u] : 0;
          const double* const row = (const double*)s_elems + lane * 16;
          int cq_next = mine ? (int)s_cqi[u] : 0;
          for (int r = 0; r < R; ++r) {
            const int cq = cq_next;
            if (r + 1 < R && mine) cq_next = s_cqi[(r + 1) * Upad + u];
            const double metric = mine ? row[cq] : 0.0;
            const bool valid = mine && need > 0;
            const int hi = valid ? __double2hiint(metric) : -1;
            const int lo = (int)((unsigned)__double2loint(metric) ^ 0x80000000u);
            const int mhi = wave_max(hi);
            const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
            const int pick = mhi >= 0 ? __ffsll((long long)__ballot(valid && hi == mhi && lo == mlo)) - 1 : -1;
            if (lane == r) owner = pick >= 0 ? c_lo + pick : -1;
            if (lane == pick) need -= G;
          }
        }
        /* sched 1 in a shape-specialised build: every lane keeps its flows (flow id 64 k + lane, k < kFK) in registers -- the
         * stage-1 reciprocal of the flow's average and an "in the race" bit -- and ranks them per RBG with the FP32 product of
         * DESIGN.md 2.6 (one multiply per flow, one wave reduction per RBG); only flows within 2^-19 of the best product can win
         * or tie, so a single survivor is the winner without any division and several are compared exactly, ascending flow id,
         * strict '>' (the reference's scan order, downlink-packet-scheduler.cpp:221-237). */
        constexpr int kFK = FIXED ? (2 * RS_JIT_U + 63) / 64 : 1;
        constexpr bool kFlowRegs = FIXED && SCHED == 1 && kFK <= 16;
        float flow_rc[kFK];
        unsigned flow_alive = 0u;
        if constexpr (kFlowRegs) {
#pragma unroll
          for (int k = 0; k < kFK; ++k) {
            const int f = 64 * k + lane, u = (f >> 1) < U ? (f >> 1) : U - 1;
            const int data = f < 2 * U ? ((f & 1) ? q_data1[u] : q_data0[u]) : 0;
            const double af = (f & 1) ? s_avgk[u] : s_avg[u];
            flow_rc[k] = __builtin_amdgcn_rcpf((float)af);
            if (data > 0) flow_alive |= 1u << k;
          }
        }
        int tbs_cap8 = 0; /* sched 1: floor(largest transport block of the table / 8): more bytes than that are never carried */
        if (SCHED == 1) {
          for (int i = lane; i < (R + 1) * 16; i += 64) tbs_cap8 = s_tbs[i] > tbs_cap8 ? s_tbs[i] : tbs_cap8;
          tbs_cap8 = wave_max(tbs_cap8) >> 3;
        }
        double run_sum = 0.0; /* sched 1, lane r: EESM sum / PRBs of the flow that holds RBG r, up to and including RBG r */
        int run_nprb = 0;
        for (int r = 0; r < R && !table7; ++r) {
          int bhi = -1, blo = (int)0x80000000, bpick = -1;
          if constexpr (kFlowRegs) {
            const float kTolF = 0x1.ffffcp-1f; /* 1 - 2^-19 */
            int cqv[kFK];
            float av[kFK];
            float best_a = 0.0f;
#pragma unroll
            for (int k = 0; k < kFK; ++k) {
              const int f = 64 * k + lane, u = (f >> 1) < U ? (f >> 1) : U - 1;
              cqv[k] = s_cqi[r * Upad + u];
            }
#pragma unroll
            for (int k = 0; k < kFK; ++k) {
              av[k] = ((flow_alive >> k) & 1u) ? s_num32[cqv[k]] * flow_rc[k] : 0.0f;
              best_a = fmaxf(best_a, av[k]);
            }
            const float mx = __int_as_float(wave_max(__float_as_int(best_a))); /* products are >= 0: they order like their bits */
            if (mx > 0.0f) {
              const float thr = mx * kTolF;
              unsigned cm = 0u;
#pragma unroll
              for (int k = 0; k < kFK; ++k) cm |= av[k] >= thr ? (1u << k) : 0u; /* thr > 0: a survivor is in the race */
              const unsigned long long holders = __ballot(cm != 0u);
              if (__popcll(holders) == 1) {
                const int src = __ffsll((long long)holders) - 1;
                const unsigned cmw = (unsigned)__builtin_amdgcn_readlane((int)cm, src);
                if ((cmw & (cmw - 1u)) == 0u) bpick = 64 * (__ffs((int)cmw) - 1) + src;
              }
              if (bpick < 0) {
                /* several flows within the tolerance (flows that were never served share one average): compared exactly, chunk
                 * by chunk in ascending flow id, strict '>' (three reductions over per-lane bests measured slower: 63.1 against
                 * 60.8 us per TTI) */
#pragma unroll
                for (int k = 0; k < kFK; ++k) {
                  if (__ballot((cm >> k) & 1u) == 0ull) continue;
                  const int f = 64 * k + lane, u = (f >> 1) < U ? (f >> 1) : U - 1;
                  bool valid = ((cm >> k) & 1u) != 0u;
                  const double metric = s_num[cqv[k]] / ((f & 1) ? s_avgk[u] : s_avg[u]); /* (se * 180000.) / the flow's own average */
                  valid = valid && metric > 0; /* the scan starts from 0 with '>' */
                  const int hi = valid ? __double2hiint(metric) : -1;
                  const int lo = (int)((unsigned)__double2loint(metric) ^ 0x80000000u);
                  const int mhi = wave_max(hi);
                  const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
                  if (mhi >= 0 && (mhi > bhi || (mhi == bhi && mlo > blo))) {
                    bhi = mhi;
                    blo = mlo;
                    bpick = 64 * k + __ffsll((long long)__ballot(valid && hi == mhi && lo == mlo)) - 1;
                  }
                }
              }
            }
          }
          for (int c0 = c_lo; c0 < c_hi && !kFlowRegs; c0 += 64) {
            const int cnd = c0 + lane;
            const int u = SCHED == 1 ? cnd >> 1 : cnd;
            bool valid = cnd < c_hi;
            double metric = 0.0;
            if (valid) {
              const int cq = s_cqi[r * Upad + u];
              if (SCHED == 1) {
                const int data = (cnd & 1) ? q_data1[u] : q_data0[u];
                valid = data > 0 && q_done[cnd] == 0;
                metric = s_num[cq] / ((cnd & 1) ? s_avgk[u] : s_avg[u]); /* (se * 180000.) / the flow's own average */
                valid = valid && metric > 0; /* the scan starts from 0 with '>' */
              } else {
                valid = (prio_in[u] & 2) != 0 && q_need[u] > 0;
                const double num = sl_eps ? s_num[cq] : 1.0, den = sl_psi ? s_avgk[u] : 1.0;
                if (!sl_custom) metric = num / den;
                else metric = (prio_in[u] & 1) == 0 ? 0.0 : hol_in[u] * num / den; /* ref: nvs :375-387 */
              }
            }
            const int hi = valid ? __double2hiint(metric) : -1;
            const int lo = (int)((unsigned)__double2loint(metric) ^ 0x80000000u);
            const int mhi = wave_max(hi);
            const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
            if (mhi >= 0 && (mhi > bhi || (mhi == bhi && mlo > blo))) {
              bhi = mhi;
              blo = mlo;
              bpick = c0 + __ffsll((long long)__ballot(valid && hi == mhi && lo == mlo)) - 1;
            }
          }
          if (lane == r) owner = bpick;
          if (bpick >= 0) {
            if (SCHED == 7) {
              if (lane == 0) q_need[bpick] -= G;
            } else {
              /* the flow's transport block so far (its PRBs in RBG order): satisfied once it carries the whole queue.  The new
               * RBG is the flow's last, so the EESM sum continues where the lane of the flow's previous RBG left it (the same
               * additions in the same order as summing all of them again) */
              /* (a flow with more data than the largest transport block of the table -- every InfiniteBuffer flow -- is never
               * satisfied: no sum to keep) */
              const int data_pick = (bpick & 1) ? q_data1[bpick >> 1] : q_data0[bpick >> 1];
              if (data_pick <= tbs_cap8) {
              const unsigned long long before = __ballot(owner == bpick && lane < r);
              const int prev = before != 0ull ? 63 - __clzll((long long)before) : 0;
              const double psum = __shfl(run_sum, prev, 64);
              const int pn = __builtin_amdgcn_readlane(run_nprb, prev);
              if (lane == r) {
                const int u = bpick >> 1;
                double sum = before != 0ull ? psum : 0.0;
                int nprb = before != 0ull ? pn : 0;
                if (per_prb) { /* the flow's per-PRB feedback (ref: downlink-packet-scheduler.cpp:245-264) */
                  const uint8_t* pr = prb_ptr(u, r);
                  for (int k = 0; k < G; ++k) sum += s_e[pr[k]];
                } else {
                  const double ev = s_e[s_cqi[r * Upad + u]];
                  for (int k = 0; k < G; ++k) sum += ev;
                }
                nprb += G;
                run_sum = sum;
                run_nprb = nprb;
                const double x = sum / (double)nprb;
                int fq = 15;
                if (!(x == 0)) {
                  fq = 1;
#pragma unroll
                  for (int k = 1; k <= 13; ++k) fq += (x <= xthr_k[k - 1]) ? 1 : 0;
                }
                if (s_tbs[(nprb / G) * 16 + fq] >= data_pick * 8) q_done[bpick] = 1;
              }
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (kFlowRegs) { /* a satisfied flow leaves the race: its lane clears the bit */
              if (q_done[bpick] != 0 && lane == (bpick & 63)) flow_alive &= ~(1u << (bpick >> 6));
            }
          }
        }
      } else if (DIRECT && (SCHED == 1 || SCHED == 7) && p.gate != nullptr) {
        /* Drop-in mode with finite queues (rs_tti_in.data_to_transmit / required_rbs): the same RBG-by-RBG race as above on the
         * caller's candidates -- flows (sched 1: every passed "user" is one flow) / the served slice's users (sched 7). */
        for (int u = lane; u < U; u += 64) {
          q_need[u] = p.gate[u];
          q_done[u] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int sl7 = (int)p.user_slice[0];
        const int sl_eps = SCHED == 7 ? (m->eps_psi[sl7] & 1) : 1, sl_psi = SCHED == 7 ? ((m->eps_psi[sl7] >> 1) & 1) : 1;
        const bool custom7 = SCHED == 7 && queue_mode_in && p.alpha[sl7] != 0;
        for (int r = 0; r < R; ++r) {
          int bhi = -1, blo = (int)0x80000000, bpick = -1;
          for (int c0 = 0; c0 < U; c0 += 64) {
            const int u = c0 + lane;
            bool valid = u < U;
            double metric = 0.0;
            if (valid) {
              const int cq = s_cqi[r * Upad + u];
              if (SCHED == 1) {
                valid = q_done[u] == 0;
                metric = s_num[cq] / s_avg[u];
                valid = valid && metric > 0;
              } else {
                valid = q_need[u] > 0;
                const double num = p.gen_exp ? p.gen_num[sl7 * 16 + cq] : (sl_eps ? s_num[cq] : 1.0);
                const double den = p.gen_exp ? s_avgk[u] : (sl_psi ? s_avgk[u] : 1.0);
                if (!custom7) metric = num / den;
                else metric = (prio_in && (prio_in[u] & 1) == 0) ? 0.0 : hol_in[u] * num / den;
              }
            }
            const int hi = valid ? __double2hiint(metric) : -1;
            const int lo = (int)((unsigned)__double2loint(metric) ^ 0x80000000u);
            const int mhi = wave_max(hi);
            const int mlo = wave_max(hi == mhi ? lo : (int)0x80000000);
            if (mhi >= 0 && (mhi > bhi || (mhi == bhi && mlo > blo))) {
              bhi = mhi;
              blo = mlo;
              bpick = c0 + __ffsll((long long)__ballot(valid && hi == mhi && lo == mlo)) - 1;
            }
          }
          if (lane == r) owner = bpick;
          if (bpick >= 0) {
            if (SCHED == 7) {
              if (lane == 0) q_need[bpick] -= G;
            } else {
              const unsigned long long mine = __ballot(owner == bpick && lane <= r);
              if (lane == r) {
                unsigned long long mm = mine;
                double sum = 0;
                int nprb = 0;
                while (mm) {
                  const int r2 = __ffsll((long long)mm) - 1;
                  mm &= mm - 1;
                  if (per_prb) {
                    const uint8_t* pr = prb_ptr(bpick, r2);
                    for (int k = 0; k < G; ++k) sum += s_e[pr[k]];
                  } else {
                    const double ev = s_e[s_cqi[r2 * Upad + bpick]];
                    for (int k = 0; k < G; ++k) sum += ev;
                  }
                  nprb += G;
                }
                const double x = sum / (double)nprb;
                int fq = 15;
                if (!(x == 0)) {
                  fq = 1;
#pragma unroll
                  for (int k = 1; k <= 13; ++k) fq += (x <= xthr_k[k - 1]) ? 1 : 0;
                }
                if (s_tbs[(nprb / G) * 16 + fq] >= q_need[bpick] * 8) q_done[bpick] = 1;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
      } else if constexpr (SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103) {
        /* DownlinkTransportScheduler's inter-slice policies (rs_interslice.h): lane r learns the slice of RBG r */
        int my_slice;
        constexpr int kS = FIXED ? RS_JIT_S : 0, kR = FIXED ? RS_JIT_R : 0;
        if constexpr (SCHED == 8) my_slice = interslice_greedy_by_row<kS, kR>(cur_rec, m, S, R, got);
        else if constexpr (SCHED == 101) my_slice = interslice_subopt<kS, kR>(cur_rec, m, (uint8_t*)(lds + o.sortx), S, R, got);
        else if constexpr (SCHED == 103) my_slice = interslice_vogel<kS, kR>(cur_rec, m, S, R, got);
        else {
#ifdef RS_STAMPS
#define RS_SCAN_ARGS s_sorted, m, S, R, got, stamp_acc
#else
#define RS_SCAN_ARGS s_sorted, m, S, R, got
#endif
          if constexpr (kVecScan) {
            if constexpr (FIXED) {
              my_slice = interslice_maximize_cell_vector<kS, kR, (kS <= 32 && kR <= 32)>(RS_SCAN_ARGS);
            } else {
              if (vec_scan && R <= 32 && S <= 32) my_slice = interslice_maximize_cell_vector<0, 0, true>(RS_SCAN_ARGS);
              else if (vec_scan) my_slice = interslice_maximize_cell_vector<0, 0, false>(RS_SCAN_ARGS);
              else my_slice = interslice_maximize_cell<0, 0>(RS_SCAN_ARGS);
            }
          } else {
            my_slice = interslice_maximize_cell<kS, kR>(RS_SCAN_ARGS);
          }
        }
        if (((kSpecSched && spec_next) || quota_next) && !p.phy_draws) {
          /* slice_rbs_offset_ is final as soon as the RBGs are dealt out (ref: :618-620): the quota wave can start TTI t+1's
           * targets while this wave still looks up the winners and adapts the links */
          if (lane < S) s_sstate[lane] = (double)(my_target - got * G);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) atomicExch(&fl_cur->greedy_done, 1);
        }
        if (lane < R && my_slice >= 0) {
          int u = cur_bu[my_slice * R + lane];
          owner = u == 0xFFFF ? -1 : u;
        }
      } else if (SCHED == 1) {
        /* ref: downlink-packet-scheduler.cpp:221-237 -- per RBG the first maximum over all flows,
         * here over the segment winners in ascending segment order */
        if (lane < R) {
          double best = 0.0;
          int sg = 0;
          for (; sg + 4 <= o.n_seg; sg += 4) { /* four segments per step: the eight LDS reads first */
            double v[4];
            int u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = s_best_metric[(sg + q) * R + lane]; u[q] = s_best_user[(sg + q) * R + lane]; }
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (u[q] != 0xFFFF && v[q] > best) { best = v[q]; owner = u[q]; }
          }
          for (; sg < o.n_seg; ++sg) {
            double v = s_best_metric[sg * R + lane];
            int u = s_best_user[sg * R + lane];
            if (u != 0xFFFF && v > best) { best = v; owner = u; }
          }
        }
      } else if (SCHED == 7 && (FIXED ? kCv.nvs_seg : p.nvs_seg) != 0) {
        /* ref: downlink-nvs-scheduler.cpp:283-297 -- per RBG the first maximum from lowest(), over the run winners in
         * ascending run order */
        if (lane < R) {
          double best = -1.7976931348623157e308;
          bool none = true;
          for (int sg = 0; sg < nvs_runs; ++sg) {
            const double v = s_best_metric[sg * R + lane];
            const int u = s_best_user[sg * R + lane];
            if (u != 0xFFFF && (none || v > best)) { best = v; owner = u; none = false; }
          }
        }
      } else {
        if (lane < R) {
          int u = s_best_user[lane];
          owner = u == 0xFFFF ? -1 : u;
        }
      }
      RS_STAMP(5);
      if (kEarly17 && SCHED == 7 && early17) {
        /* the winner tables are in this wave's registers now: the scanning waves may overwrite them with TTI t+1's */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicExch(&fl_cur->greedy_done, 1);
      }

      /* ---------------- P5: link adaptation + DoStopSchedule counters (lanes = RBGs) ---------------- */
      /* lanes holding the same user; the lowest one (leader) handles the user */
      constexpr bool kFlows = kQSerial && SCHED == 1; /* owner = flow id 2 * user + bearer */
      /* owner + 1 < 4096 = two base-64 digits: the lanes that share both digits share the owner.  One LDS atomic OR per digit
       * into the sort's two 64-entry mask arrays (idle here), ~15 instructions and one LDS round trip */
      unsigned long long same = 0ull;
      {
        const bool has = lane < R && owner >= 0;
        const int key = has ? owner + 1 : 0;
        m->maskA[lane] = 0ull;
        m->maskB[lane] = 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (has) {
          atomicOr(&m->maskA[key & 63], 1ull << lane);
          atomicOr(&m->maskB[key >> 6], 1ull << lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const unsigned long long lo = m->maskA[key & 63], hi = m->maskB[key >> 6];
        if (has) same = lo & hi;
      }
      const bool leader = owner >= 0 && (same & ((1ull << lane) - 1ull)) == 0;
      const unsigned long long lead_mask = __ballot(leader);
      served_prev = __popcll(lead_mask);
      /* ref: :618-620 slice_rbs_offset_ = target - final_rbgs*rbg_size */
      if (kTransport && lane < S && !(QUEUE && *q_any == 0))
        s_sstate[lane] = (double)(((kSpecSched || kHoldSched) ? my_target : m->target[lane]) - got * G);
      if (lane == 0) m->served = served_prev;
      if (kHoldSched) {
        if (leader) atomicOr(&hold_served[(owner & 2047) >> 5], 1u << (owner & 31)); /* next TTI: these users' items are scanned again */
        if (quota_next) { /* slice offsets and the served count are in place: the quota wave may finish TTI t+1's quotas */
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) atomicExch(&fl_cur->greedy_done, 2);
        }
      }
      if (kSpecSched && spec_next) {
        /* the allocation is decided: tell the scanning waves who was served (their speculative winners among these need a
         * second look) and let the quota wave start TTI t+1's quotas (slice offsets and the served count are in place) */
        if (leader) atomicOr(&served_bits[owner >> 5], 1u << (owner & 31));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicExch(&fl_cur->greedy_done, 2); /* 1: slice offsets final, 2: served set published too */
      }
      int tbs = 0, nprb = 0, fcqi = 0, mcs = 0;
      int tbs_bytes_next = 0; /* bytes of this TTI's grant, for the served user's next EWMA update (speculated TTIs) */
      {
        /* ref: :638-651 -- PRBs in RBG-ascending order, G identical adds per RBG
         * (src/utility/eesm-effective-sinr.h:33-46 with the exp() values tabulated by the host) */
        unsigned long long mm = leader ? same : 0ull;
        double sum = 0;
        const uint8_t* col = s_cqi + (owner < 0 ? 0 : (kFlows ? owner >> 1 : owner));
        /* every RBG lane looks up the E value of its own RBG for its owner once (two dependent LDS reads, all lanes side by
         * side); the leaders then collect their lanes' values in RBG order with lane reads instead of two LDS round trips per RBG */
        if (!per_prb) {
          const double ev_mine = (lane < R && owner >= 0) ? s_e[col[lane * Upad]] : 0.0;
          while (__ballot(mm != 0ull) != 0ull) { /* every lane stays in the loop: a lane read from a masked-off lane returns 0 */
            const bool more = mm != 0ull;
            const int r2 = more ? __ffsll((long long)mm) - 1 : 0;
            mm &= mm - 1;
            const double ev = __shfl(ev_mine, r2, 64);
            if (more) {
              for (int k = 0; k < G; ++k) sum += ev;
              nprb += G;
            }
          }
        }
        while (mm) {
          const int r2 = __ffsll((long long)mm) - 1;
          mm &= mm - 1;
          if (per_prb) {
            /* per-PRB reports (the simulated channel's, or a per-PRB batch source): read the RBG's PRBs from HBM */
            const uint8_t* pr = prb_ptr(kFlows ? owner >> 1 : owner, r2);
            for (int k = 0; k < G; ++k) sum += s_e[pr[k]];
          } else {
            const double ev = s_e[col[r2 * Upad]];
            for (int k = 0; k < G; ++k) sum += ev;
          }
          nprb += G;
        }
        /* the synthetic-experiment build (rs_config.synthetic_exp; ref: :653-659, nvs :336-342): the transport block adds up every
         * allocated PRB at the MCS of its own CQI.  Every RBG lane works out its RBG's bits, the leaders add up their lanes
         * (integers: any order). */
        int syn_bits = 0;
        if (p.synthetic && (SCHED == 7 || kTransport)) {
          int t1 = 0;
          if (lane < R && owner >= 0) {
            if (per_prb) {
              const uint8_t* pr = prb_ptr(owner, lane);
              for (int k = 0; k < G; ++k) t1 += m->tbs1_of_cqi[pr[k]];
            } else {
              t1 = G * m->tbs1_of_cqi[col[lane * Upad]];
            }
          }
          unsigned long long ms = leader ? same : 0ull;
          while (__ballot(ms != 0ull) != 0ull) {
            const bool more = ms != 0ull;
            const int r2 = more ? __ffsll((long long)ms) - 1 : 0;
            ms &= ms - 1;
            const int v = __shfl(t1, r2, 64);
            if (more) syn_bits += v;
          }
        }
        if (leader) {
          const double x = sum / (double)nprb;
          if (x == 0) {
            fcqi = 15;
          } else {
            fcqi = 1;
#pragma unroll
            for (int k = 1; k <= 13; ++k) fcqi += (x <= xthr_k[k - 1]) ? 1 : 0; /* thresholds: wave-uniform, read before the TTI loop */
          }
          mcs = m->mcs_of_cqi[fcqi];
          tbs = (p.synthetic && (SCHED == 7 || kTransport)) ? syn_bits : s_tbs[(nprb / G) * 16 + fcqi];
          /* DoStopSchedule, ref: :170-221 (bytes = bits/8, capped by dataToTransmit = 1e8) */
          int bytes = tbs / 8;
          if (bytes > 100000000) bytes = 100000000;
          if (bytes > 0) {
            if (kFlows) {
              /* this flow's transport block: credited to its own bearer by the owner thread (stop_schedule_user) */
              if (owner & 1) q_grant1[owner >> 1] = bytes | (nprb << RS_TX_NPRB_SHIFT);
              else s_tx[owner >> 1] = bytes | (nprb << RS_TX_NPRB_SHIFT);
            } else if (kCumRegs || QUEUE) {
              /* the owner thread of P1 counts it (registers) / splits it over the user's bearers and dequeues (queue model) */
              atomicAdd(&s_tx[owner], bytes | (nprb << RS_TX_NPRB_SHIFT)); /* one ds_add, nothing to wait for (this lane is the word's only writer) */
            } else {
              /* the next EWMA update consumes the bytes: in a speculated TTI that happens right below, on this lane */
              if (!(kSpecSched && spec_next)) s_tx[owner] += bytes;
              /* RadioBearer::m_cumulativeBytes / m_cumulativeRBs live in HBM: fire-and-forget atomics */
              atomicAdd((unsigned long long*)&p.cum_bytes[(size_t)cell * U + owner], (unsigned long long)bytes);
              atomicAdd((unsigned long long*)&p.cum_rbs[(size_t)cell * U + owner], (unsigned long long)nprb);
            }
          }
          if ((kSpecSched && spec_next) || (kHoldSched && ewma_next) || early17) tbs_bytes_next = bytes;
        }
      }
      if ((kHoldSched && ewma_next) || early17) {
        /* the exact EWMA of the served users for TTI t+1, on top of the decay the other waves applied (all of them first).
         * (Handing these ~900 cycles to wave 1 through an LDS list was measured and lost -- NVS 164 against 184 M TTIs/s, GreedyByRow
         * 97.8 against 98.4: the update is a dependent chain that nothing overlaps with, whoever runs it, and the hand-over adds a
         * polling round trip; profiles/r04_sched17.md) */
        while (rs_lds_load(&fl_cur->ctr_p1) < nwaves - 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (leader) {
          const double t_next = t + 0.001;
          const double dt_next = t_next - t; /* Now - m_lastUpdate of the next update: m_lastUpdate is this TTI's time */
          double a = SCHED == 1 ? s_avgk[owner] : s_avg[owner];
          const double rate = (double)(tbs_bytes_next * 8) / dt_next;
          const double beta = 0.02;
          a = a + (beta * rate);
          if (a < 1) a = 1;
          s_avg[owner] = a;
          if (SCHED == 1) s_rcp32[owner] = __builtin_amdgcn_rcpf((float)a);
          else pf_terms(owner, a);
        }
      }
      if (kSpecSched && spec_next) {
        /* Exact EWMA of the served users for TTI t+1 (ref: src/flows/radio-bearer.cpp:139-164): the scanning waves left
         * (1 - beta) * avg, unclamped, in s_avg; adding beta * rate gives the reference's sum of the two rounded products.
         * Not before every speculative scan is over: they must all have seen one state, the speculative one. */
        RS_STAMP(7);
        while (rs_lds_load(&fl_cur->ctr_p3) < nwaves - 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        RS_STAMP(6);
        if (leader) {
          const double t_next = t + 0.001;
          const double dt_next = t_next - t; /* Now - m_lastUpdate of the next update: m_lastUpdate is this TTI's time */
          double a = s_avg[owner];
          const double rate = (double)(tbs_bytes_next * 8) / dt_next;
          const double beta = 0.02;
          a = a + (beta * rate);
          if (a < 1) a = 1;
          s_avg[owner] = a;
          pf_terms(owner, a);
        }
      }
      /* optional log */
      if (p.log_map) {
        size_t row = (size_t)cell * p.n_ttis + tti;
        if (lane < R) p.log_map[row * R + lane] = (int16_t)owner;
        if (lane < S) {
          if (p.log_quota) p.log_quota[row * S + lane] = (int16_t)((kSpecSched || kHoldSched) ? my_quota : m->quota[lane]);
          if (p.log_target) p.log_target[row * S + lane] = (int16_t)((kSpecSched || kHoldSched) ? my_target : m->target[lane]);
        }
        if (leader) {
          if (kFlows) { /* two flows of one user may both hold RBGs: the user's row shows their sum */
            if (p.log_tbs) atomicAdd(&p.log_tbs[row * U + (owner >> 1)], tbs);
          } else {
            if (p.log_tbs) p.log_tbs[row * U + owner] = tbs;
            if (p.log_uinfo) p.log_uinfo[row * U + owner] = nprb | (fcqi << 16) | (mcs << 24);
          }
        }
      }
    }
    if (kEarly17 && SCHED == 7 && early17 && wave == quota_wave) {
      /* SelectSliceToServe of TTI t+1 first: the scanning waves wait for it */
      nvs_pick(nvs_word_nxt);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicExch(&fl_cur->n_fix, 1);
    }
    if (((kHoldSched && ewma_next) || early17) && wave != 0) {
      const int nsp = nt - 64, me = tid - 64;
      for (int u = me; u < U; u += nsp) {
        double a = s_avg[u];
        if (a < 1) a = 1;
        const double beta = 0.02;
        const double us = (1 - beta) * a;
        if (SCHED == 1) {
          /* the per-flow PF scheduler divides by s_avg itself: it keeps the clamped value, the raw product waits in s_avgk (unused
           * by this scheduler) for wave 0 */
          s_avgk[u] = us;
          s_avg[u] = us < 1 ? 1.0 : us;
          s_rcp32[u] = __builtin_amdgcn_rcpf((float)(us < 1 ? 1.0 : us));
        } else {
          s_avg[u] = us; /* unclamped: a served user's exact update adds beta * rate to it */
          pf_terms(u, us < 1 ? 1.0 : us);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&fl_cur->ctr_p1, 1);
    }
    if (kEarly17 && SCHED == 7 && early17) {
      /* TTI t+1's slice is known once the quota wave has published it; its metric scan can run now when that slice is not the one
       * wave 0 is serving (then no average it reads changes any more) and TTI t+1 reads this CQI grid */
      if (wave != 0) {
        while (rs_lds_load(&fl_cur->n_fix) < 1) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int seg_next = rs_lds_load(nvs_word_nxt);
        if (early_scan_ok && seg_next != seg_this) {
          /* every decayed average first (the scan reads other threads' users), and wave 0 must have read TTI t's winners */
          while (rs_lds_load(&fl_cur->ctr_p1) < nwaves - 1) __builtin_amdgcn_s_sleep(1);
          while (rs_lds_load(&fl_cur->greedy_done) < 1) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          seg_lo = seg_next; /* (this wave's copies: wave 0 keeps TTI t's; everybody re-reads them at the top of TTI t+1) */
          int n_next = o.n_items;
          if (nvs_split) {
            nvs_lo = m->seg_begin[seg_lo];
            nvs_hi = m->seg_begin[seg_lo + 1];
            nvs_first = nvs_lo & ~7;
            nvs_runs = idiv_small(nvs_hi - nvs_first + nvs_seg - 1, nvs_seg);
            n_next = R * nvs_runs;
          }
          const int nsp = nt - 64, me = tid - 64;
          for (int it = me; it < n_next; it += nsp) scan_item(it, cur_bu, cur_rec, RsInt<0>{});
        }
      }
    }
    if (quota_next && wave == quota_wave) {
      /* TTI t+1's draws and remainder rotations need nothing of TTI t (unless the error model's draws, one per UE served, come
       * first on the shared stream); its targets need the slice offsets wave 0 publishes with the allocation */
      if (!p.phy_draws) quota_draws(0);
      const int need_stage = p.phy_draws ? 2 : 1;
      while (rs_lds_load(&fl_cur->greedy_done) < need_stage) __builtin_amdgcn_s_sleep(RS_SPEC_NAP);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (p.phy_draws) quota_draws(rs_lds_load(&m->served));
      quota_targets();
    }
    if (kHoldSched && ewma_next && wave != 0 && hold_ok && n_items_rt <= 64 * nwaves) {
      /* my list for TTI t+1 (the held bits, the winners and -- once wave 0 has published it -- the served set are what the top
       * of TTI t+1 would read; if that TTI turns out to scan everything, the list is simply not used) */
      while (rs_lds_load(&fl_cur->greedy_done) < 2) __builtin_amdgcn_s_sleep(RS_SPEC_NAP);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      pre_listed = hold_pack(0, wave);
      /* wave 0 is busy until the TTI ends: wave 1 packs its list too (the count travels in an LDS word) */
      if (wave == 1) {
        const int n0 = hold_pack(0, 0);
        if (lane == 0) m->pad[1] = n0;
      }
    }
    if (kSpecSched && spec_next && wave != 0) {
      /* ---------------- the other waves meanwhile: TTI t+1 as if nobody were served in TTI t ---------------- */
      const int nsp = nt - 64, me = tid - 64; /* scanning threads and my index among them */
#if defined(RS_STAMPS) && defined(RS_STAMPS_W1)
      stamp1_prev = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_setprio(RS_SPEC_PRIO);
      /* P1: avg' = (1 - beta) * avg + beta * 0 = (1 - beta) * avg exactly; the unclamped product stays in s_avg (a served
       * user's exact update adds beta * rate to it), the PF terms use the clamped value */
      for (int u = me; u < U; u += nsp) {
        double a = s_avg[u];
        if (a < 1) a = 1;
        const double beta = 0.02;
        const double us = (1 - beta) * a;
        s_avg[u] = us;
        pf_terms(u, us < 1 ? 1.0 : us);
      }
      RS_STAMP1(0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&fl_cur->ctr_p1, 1);
      while (rs_lds_load(&fl_cur->ctr_p1) < nwaves - 1) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      RS_STAMP1(1);
      /* P3 on the speculative state */
      const int n_spec = spec_items(n_items);
      for (int it = me; it < n_spec; it += nsp) scan_item(it, nxt_bu, nxt_rec, RsInt<0>{});
      /* TTI t+1's draws and remainder rotations need nothing of TTI t (unless the error model's draws, one per UE served,
       * come first on the shared stream) */
      if (wave == quota_wave && !p.phy_draws) quota_draws(0);
      /* which of my winners were served?  (wave 0 publishes the served set as soon as the allocation is decided) */
      RS_STAMP1(2);
      if (wave == quota_wave) {
        /* TTI t+1's quotas as soon as TTI t's slice offsets are final (with the error model's draws on the stream: once the
         * served count is known too); wave 0 will wait for this wave, so it runs at wave 0's priority from here on */
        __builtin_amdgcn_s_setprio(3);
        const int need_stage = p.phy_draws ? 2 : 1;
        while (rs_lds_load(&fl_cur->greedy_done) < need_stage) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (p.phy_draws) quota_draws(rs_lds_load(&m->served));
        quota_targets();
        RS_STAMP1(5);
      }
      while (rs_lds_load(&fl_cur->greedy_done) < 2) __builtin_amdgcn_s_sleep(RS_SPEC_NAP);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      RS_STAMP1(3);
      for (int it0 = (wave - 1) * 64; it0 < n_spec; it0 += nsp) {
        const int it = it0 + lane;
        bool need = false;
        if (it < n_spec) {
          const int w = nxt_bu[it];
          need = w != 0xFFFF && ((served_bits[w >> 5] >> (w & 31)) & 1u) != 0u;
        }
        const unsigned long long mk = __ballot(need);
        if (mk != 0ull) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&fl_cur->n_fix, __popcll(mk));
          base = __builtin_amdgcn_readfirstlane(base);
          const int slot = base + __popcll(mk & ((1ull << lane) - 1ull));
          if (need && slot < RS_FIX_CAP) fix_list[slot] = (uint16_t)it;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&fl_cur->ctr_p3, 1);
      RS_STAMP1(4);
    }
    __builtin_amdgcn_s_setprio(0);
    RS_STAMP(7);
    __syncthreads();
    RS_STAMP(8);
    have_spec = kSpecSched && spec_next;
    have_quota = quota_next || (kEarly17 && SCHED == 7 && early17);
    have_ewma = (kHoldSched && ewma_next) || early17;
    if (kEarly17 && SCHED == 7) {
      /* did the scanning waves find TTI t+1's winners?  The same test they made, on values that are final behind the barrier */
      have_scan = early17 && early_scan_ok && rs_lds_load(nvs_word_nxt) != seg_this;
    }
    /* (the packing condition is the same on every wave: wave 0 learns here that wave 1 packed its list) */
    if (wave == 0 && kHoldSched && ewma_next && hold_ok && n_items_rt <= 64 * nwaves) {
      pre_listed = rs_lds_load(&m->pad[1]);
    }
    served_prev = m->served;
    n_done += 1;
    if (++epoch_pos == p.refresh) {
      epoch_pos = 0;
      if (++epoch == p.n_epochs && p.epoch_wrap) epoch = 0;
    }
    if (!kDirect) t += 0.001; /* ref: src/core/eventScheduler/simulator.cc:117-126 */
  }

  if constexpr (QUEUE) {
    /* DoStopSchedule of the launch's last TTI, so that the bearers' counters and queues the host reads are complete */
    for (int u = tid; u < U; u += nt) stop_schedule_user(u);
    if (q_lds) { /* the bearers' words back to HBM, by their owner threads */
      const size_t n = (size_t)p.n_cells * 2 * U;
      for (int u = tid; u < U; u += nt)
        for (int b = 0; b < 2; ++b) {
          const size_t bi = bearer_index(u, b);
          p.b_avg[bi] = qs_avg[b * U + u];
#pragma unroll
          for (int f = 0; f < 7; ++f) p.q_head[(size_t)f * n + bi] = qs_i[(f * 2 + b) * U + u];
        }
    }
  }
  /* ---------------- store the cell ---------------- */
#pragma unroll
  for (int ku = 0; ku < (kCumRegs ? kKU : 1); ++ku) {
    if (!kCumRegs) break;
    const int u = tid + ku * nt;
    if (u < U) {
      /* totals = what the EWMA updates consumed + the last TTI's service still waiting in s_tx (marked as counted) */
      const int v = s_tx[u];
      const long long b = (long long)cum_b[ku] + (v & RS_TX_BYTES_MASK), r = (long long)cum_r[ku] + ((v >> RS_TX_NPRB_SHIFT) & RS_TX_NPRB_MASK);
      if (b != 0) p.cum_bytes[(size_t)cell * U + u] += b;
      if (r != 0) p.cum_rbs[(size_t)cell * U + u] += r;
      p.avg[(size_t)cell * U + u] = ((kSpecSched || kHoldSched || kEarly17) && s_avg[u] < 1) ? 1.0 : s_avg[u];
      p.tx_bytes[(size_t)cell * U + u] = v ? (v | RS_TX_COUNTED) : 0;
    }
  }
  for (int u = tid; u < U && !kCumRegs; u += nt) {
    p.avg[(size_t)cell * U + u] = ((kSpecSched || kHoldSched) && s_avg[u] < 1) ? 1.0 : s_avg[u];
    p.tx_bytes[(size_t)cell * U + u] = s_tx[u];
  }
  if (tid < S) p.slice_state[(size_t)cell * S + tid] = s_sstate[tid];
  if (wave == quota_wave && lane < 31) scal->rng_r[lane] = rng.r;
  if (tid == 0) {
    scal->t = t;
    scal->last_update = last_update;
    scal->last_sent = last_sent;
    scal->reported = reported;
    scal->cqi_row = cqi_row;
    scal->served_prev = served_prev;
    scal->n_done = n_done;
    if (local_err) atomicExch(p.err, local_err);
#ifdef RS_STAMPS
    if (p.stamps) {
      for (int i = 0; i < 12; ++i) p.stamps[(size_t)cell * 20 + i] = stamp_acc[i];
#ifndef RS_STAMPS_W1
      for (int i = 0; i < 8; ++i) p.stamps[(size_t)cell * 20 + 12 + i] = sort_sub[i];
#endif
    }
#endif
  }
#if defined(RS_STAMPS) && defined(RS_STAMPS_W1)
  if (tid == RS_STAMPS_W1_TID && p.stamps)
    for (int i = 0; i < 8; ++i) p.stamps[(size_t)cell * 20 + 12 + i] = sort_sub[i];
#endif
  if (wave == quota_wave && lane == 0) {
    scal->rng_f = rng.f;
    scal->rng_b = rng.b;
  }
}

}  // namespace

#ifndef RS_JIT_BUILD
template <int SCHED, int EPT, bool DIRECT, bool QUEUE = false>
__global__ void __launch_bounds__(512, 4) rs_cell_kernel(RsLaunch p) {
  extern __shared__ __align__(16) unsigned char lds[];
  rs_cell_body<SCHED, EPT, false, DIRECT, QUEUE>(p, lds);
}
#else
#ifndef RS_JIT_DIRECT
#define RS_JIT_DIRECT 0 /* 1: the drop-in entry point's one-TTI form (rs_ctx_specialize) */
#endif
/* shape-specialised entry point compiled at run time (rs_jit.cpp): static LDS of exactly the carve's size.  RS_JIT_WPE = waves
 * per SIMD the register allocation must leave room for (4: 128 VGPRs, two 512-thread cells per CU; rs_jit.cpp passes 5 for the
 * 640-thread cells of the 64-RBG grid: two cells = 20 waves per CU) */
#ifndef RS_JIT_WPE
#define RS_JIT_WPE 4
#endif
extern "C" __global__ void __launch_bounds__(RS_JIT_NT, RS_JIT_WPE) rs_cell_kernel_jit(RsLaunch p) {
  constexpr RsCarve kCv = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ);
  __shared__ __align__(16) unsigned char lds[kCv.lds_bytes];
  constexpr int kEpt = (RS_JIT_SCHED != 9 && RS_JIT_SCHED != 10) ? 0 : (kCv.ept <= 4 ? kCv.ept : 0);
  rs_cell_body<RS_JIT_SCHED, kEpt, true, RS_JIT_DIRECT != 0, (RS_JIT_CARVEQ >= 2)>(p, lds);
}
#endif

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
                                    int R, uint64_t seed, int64_t first_cell, RsCdf cdf) {
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
        /* keyed by (seed, GLOBAL cell id, epoch, user, rbg): a cell's grids do not depend on how the cells are sharded over
         * ranks nor on how many epochs were generated */
        const uint64_t gcell = (uint64_t)(first_cell + g / n_epochs), ep = (uint64_t)(g % n_epochs);
        uint64_t h = splitmix64(seed ^ splitmix64(splitmix64(gcell * 0x100000001B3ull + ep) + (uint64_t)idx));
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

/* measurement only: 16 bytes per lane streaming copy, to quote the attainable HBM rate next to the 8 TB/s spec
 * (SURVEY 8d).  Grid-stride so that a launch of a few workgroups per CU covers any size. */
__global__ void __launch_bounds__(256) rs_copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
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

#if !defined(__HIPCC_RTC__) && !defined(RS_JIT_BUILD)
/* host-callable launchers (defined here so that the kernels stay in one translation unit) */
extern "C" hipError_t rs_launch_cells(const RsLaunch* p, int threads, hipStream_t stream) {
  dim3 grid(p->n_cells), block(threads);
  const int N = p->R * p->S;
  const int ept = (N + threads - 1) / threads;
  /* batches and the drop-in entry point (p->direct) run different instantiations */
#define RS_LAUNCH_CELL(SCHED_, EPT_)                                                                             \
  do {                                                                                                           \
    if (p->direct) hipLaunchKernelGGL((rs_cell_kernel<SCHED_, EPT_, true>), grid, block, p->lds_bytes, stream, *p); \
    else hipLaunchKernelGGL((rs_cell_kernel<SCHED_, EPT_, false>), grid, block, p->lds_bytes, stream, *p);         \
  } while (0)
#define RS_LAUNCH_QUEUE(SCHED_, EPT_) \
  hipLaunchKernelGGL((rs_cell_kernel<SCHED_, EPT_, false, true>), grid, block, p->lds_bytes, stream, *p)
  if (p->bearer_kind != nullptr) { /* finite queues (batches of the transport schedulers only; the host has checked) */
    switch (p->sched) {
      case 1: RS_LAUNCH_QUEUE(1, 0); break;
      case 7: RS_LAUNCH_QUEUE(7, 0); break;
      case 8: RS_LAUNCH_QUEUE(8, 0); break;
      case 101: RS_LAUNCH_QUEUE(101, 0); break;
      case 103: RS_LAUNCH_QUEUE(103, 0); break;
      case 9:
        if (ept <= 1) RS_LAUNCH_QUEUE(9, 1);
        else if (ept <= 2) RS_LAUNCH_QUEUE(9, 2);
        else if (ept <= 3) RS_LAUNCH_QUEUE(9, 3);
        else if (ept <= 4) RS_LAUNCH_QUEUE(9, 4);
        else RS_LAUNCH_QUEUE(9, 0);
        break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (p->sched) {
    case 1: RS_LAUNCH_CELL(1, 0); break;
    case 7: RS_LAUNCH_CELL(7, 0); break;
    case 8: RS_LAUNCH_CELL(8, 0); break;
    case 101: RS_LAUNCH_CELL(101, 0); break;
    case 103: RS_LAUNCH_CELL(103, 0); break;
    case 11: RS_LAUNCH_CELL(11, 0); break;
    case 10:
      if (ept <= 1) RS_LAUNCH_CELL(10, 1);
      else if (ept <= 2) RS_LAUNCH_CELL(10, 2);
      else if (ept <= 3) RS_LAUNCH_CELL(10, 3);
      else if (ept <= 4) RS_LAUNCH_CELL(10, 4);
      else return hipErrorInvalidValue;
      break;
    case 9:
      if (ept <= 1) RS_LAUNCH_CELL(9, 1);
      else if (ept <= 2) RS_LAUNCH_CELL(9, 2);
      else if (ept <= 3) RS_LAUNCH_CELL(9, 3);
      else if (ept <= 4) RS_LAUNCH_CELL(9, 4);
      else RS_LAUNCH_CELL(9, 0);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

extern "C" hipError_t rs_prepare_kernels(int max_lds_bytes) {
#define RS_BOTH(SCHED_, EPT_) (const void*)rs_cell_kernel<SCHED_, EPT_, false>, (const void*)rs_cell_kernel<SCHED_, EPT_, true>
  const void* fns[] = {RS_BOTH(1, 0),  RS_BOTH(7, 0),  RS_BOTH(8, 0),  RS_BOTH(101, 0), RS_BOTH(103, 0), RS_BOTH(11, 0),
                       RS_BOTH(10, 1), RS_BOTH(10, 2), RS_BOTH(10, 3), RS_BOTH(10, 4),
                       RS_BOTH(9, 0),  RS_BOTH(9, 1),  RS_BOTH(9, 2),  RS_BOTH(9, 3),  RS_BOTH(9, 4),
                       (const void*)rs_cell_kernel<1, 0, false, true>, (const void*)rs_cell_kernel<7, 0, false, true>,
                       (const void*)rs_cell_kernel<8, 0, false, true>, (const void*)rs_cell_kernel<101, 0, false, true>,
                       (const void*)rs_cell_kernel<103, 0, false, true>, (const void*)rs_cell_kernel<9, 0, false, true>,
                       (const void*)rs_cell_kernel<9, 1, false, true>, (const void*)rs_cell_kernel<9, 2, false, true>,
                       (const void*)rs_cell_kernel<9, 3, false, true>, (const void*)rs_cell_kernel<9, 4, false, true>};
#undef RS_BOTH

  for (const void* f : fns) {
    hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds_bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

extern "C" hipError_t rs_launch_synth(uint8_t* epochs, int64_t grid_stride, int n_cells, int n_epochs, int U, int R,
                                      uint64_t seed, int64_t first_cell, const uint32_t* cdf16, hipStream_t stream) {
  RsCdf cdf;
  for (int i = 0; i < 16; ++i) cdf.c[i] = cdf16[i];
  int64_t total = (int64_t)n_cells * n_epochs * (grid_stride >> 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rs_synth_cqi_kernel, dim3(blocks), dim3(256), 0, stream, epochs, grid_stride, n_cells, n_epochs,
                     U, R, seed, first_cell, cdf);
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

extern "C" hipError_t rs_launch_copy_probe(const void* src, void* dst, size_t bytes, hipStream_t stream) {
  hipLaunchKernelGGL(rs_copy_probe_kernel, dim3(256 * 4), dim3(256), 0, stream, (const uint4*)src, (uint4*)dst, bytes >> 4);
  return hipGetLastError();
}

#endif /* host launchers */
