/*
 * rs_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the RadioSaber downlink RBG allocation path.
 *
 * One workgroup = one cell.  The workgroup keeps the cell's whole scheduling state in LDS (RBG-major
 * CQI grid u8[R][Upad], PF averages f64[U], the CQI->rate / EESM / TBS tables, slice quotas) and runs
 * n_ttis complete DoSchedule() iterations back to back:
 *
 *   P0  CQI refresh (every 40 TTIs)        HBM -> LDS, a straight 16-byte copy per lane (the device-resident grids are stored as
 *                                          the LDS image; the drop-in entry point transposes the caller's [U][R] block)
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
 * How the file reads (round 4): rs_cell_body below is the skeleton -- mode switches, the LDS carve's pointers, "load the cell", the TTI loop,
 * "store the cell"; the phases of the loop are textual fragments included inside the function, in execution order:
 *   rs_phase_queues.inc       queue-model helpers (finite MAC queues; used when QUEUE)
 *   rs_phase_p0_p1.inc        P0 + P1          rs_phase_p2.inc   P2          rs_phase_nvs_sampler.inc   scheduler 11
 *   rs_phase_p3.inc           P3 (scan_item, held winners, fix-up of a speculated TTI)
 *   rs_phase_p4.inc           P4 on all waves (sort emulation, UpperBound)
 *   rs_phase_p4_serial.inc    wave 0: the inter-slice policy / per-RBG reduction      rs_phase_p5.inc   wave 0: P5
 *   rs_phase_next.inc         the other waves meanwhile: TTI t+1 prepared (kEarly17, held winners, speculation)
 * One function on purpose: the per-thread state (sort records, byte counters, quota lanes) lives in registers across phases.
 *
 * Template parameters of the cell body: SCHED = the reference's CLI scheduler number (1, 7, 8, 9, 10, 11; 101 = SubOpt, 103 = Vogel),
 * EPT = sort positions per thread (0: state in LDS, any size), FIXED = shape-specialised build, DIRECT = the drop-in
 * entry point's one-TTI form on caller-provided state.  Wave-level building blocks live in rs_wave.h.
 *
 * The same source is compiled three ways: into the library with the cell shape as launch arguments; at run time (hiprtc,
 * rs_jit.cpp) with the shape as compile-time constants (RS_JIT_*); and as the LEAN build of that kernel (RS_JIT_LEAN), in which the
 * launch's unused run-time options are constants too (rs_cell_kernel_jit below).
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
/* Issue priorities with the co-resident cell in mind (round 5).  Two cells share a CU; at equal s_setprio the SIMDs' arbiters prefer
 * the OLDER wave, so the cell that was dispatched first wins every tie for the whole launch: of the headline batch's 512 cells the
 * first 256 finish an 8 000-TTI launch after 105.5 ms, the second 256 after 114.7 -- alone on their CUs for the last 9 ms
 * (tools/cell_spread.py).  RS_SETPRIO(x) raises the level by one during the cell's own windows of the chip-wide 100 MHz clock
 * (s_memrealtime, bit RS_PRIO_WINDOW_LOG2): even windows for cells of even dispatch round (blockIdx.x / compute units), odd windows
 * for the others, so that each of two co-resident cells is preferred half of the time whatever their ages. */
#ifndef RS_PRIO_WINDOW_LOG2
#define RS_PRIO_WINDOW_LOG2 15 /* 2^15 ticks of 10 ns = 0.33 ms, about twenty TTIs */
#endif
#define RS_SETPRIO(x)                                                        \
  do {                                                                       \
    if (prio_boost) __builtin_amdgcn_s_setprio((x) < 3 ? (x) + 1 : 3);       \
    else __builtin_amdgcn_s_setprio(x);                                      \
  } while (0)
#define RS_SPEC_NAP 2    /* s_sleep argument (x 64 cycles) while the scanning waves wait for the allocation */
#define RS_SERIAL_PRIO 3 /* issue priority of the wave that runs the serial end of the TTI (inter-slice policy, link adaptation) */
#ifndef RS_SPEC_PRIO
#define RS_SPEC_PRIO 0   /* issue priority of the scanning waves during the serial phase */
#endif
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
#ifdef RS_STAMPS
  const unsigned long long stamp_entry = __builtin_readcyclecounter(); /* diagnostic build, one-TTI kernels: the load phase is slot 9, the store phase slot 10 */
#endif
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
  constexpr RsCarve kCvQ = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ, RS_JIT_WIN);
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
  constexpr RsCarve kCv = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ, RS_JIT_WIN);
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
#ifndef RS_VEC_MAX_R
#define RS_VEC_MAX_R 32
#endif
  constexpr int kVecMaxR = RS_VEC_MAX_R;
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
  /* Per-flow PF over backlogged users (shape-specialised batches): its arg-max has no slices -- per RBG the first maximum over ALL
   * users -- so the scan can be dealt out by RBG instead of by (32-user run, RBG) item: every wave takes RBGs nwaves - 1 - wave,
   * + nwaves, ... with all users in its lanes (8 or 16 consecutive ones per lane, their stage-1 reciprocals in registers), ranks them
   * with the FP32 product of DESIGN.md 3.2, reduces over the wave (DPP) and settles ties exactly: an RBG's winner is final when its
   * wave is done with it -- no per-item exact metric, no reduction over run winners on wave 0 (rs_phase_p3.inc).
   * The scan is VALU-bound either way (five instructions per (user, RBG) product on every SIMD of the CU); what this form saves is
   * the second stage.  Same-box A/B, 512 cells (tools/experiments/r04/run17.sh): 1 000 UEs x 25 RBGs 112.7 against 96.5 M TTIs/s, 500 x 25 161.5
   * against 163.3, 500 x 64 96.7 against 96.3 -- so: from 16 users per lane on (-DRS_PF1_ALWAYS: every shape). */
#if defined(RS_NO_PF1_LANES)
  constexpr bool kPf1 = false;
#elif defined(RS_PF1_ALWAYS)
  constexpr bool kPf1 = FIXED && !DIRECT && !QUEUE && SCHED == 1 && RS_JIT_U <= 2048;
#else
  constexpr bool kPf1 = FIXED && !DIRECT && !QUEUE && SCHED == 1 && RS_JIT_U > 512 && RS_JIT_U <= 2048;
#endif
  /* NVS scans the next TTI's slice four lanes per item (quad_scan, rs_phase_p3.inc) when its slices are scanned whole (no split
   * runs) and the batch's longest slice window is known and fits two groups of eight users per lane */
  constexpr bool kQuad7 = kEarly17 && SCHED == 7 && kCv.nvs_seg == 0 && RS_JIT_WIN > 0 && RS_JIT_WIN <= 64;

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
  if (!DIRECT && tid == 0) { /* (stored at once: nothing stays live across the launch) */
    scal->clk_begin = __builtin_readcyclecounter();
    scal->real_begin = __builtin_amdgcn_s_memrealtime();
  }

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
  /* (grid words per thread: the caller's [U][R] block, or the context's device-resident image of it -- [R][Upad], a little larger) */
  constexpr int kPGsrc = (FIXED && DIRECT) ? ((RS_JIT_U * RS_JIT_R + 15) / 16 + RS_JIT_NT - 1) / RS_JIT_NT : 1;
  constexpr int kPGimg = (FIXED && DIRECT) ? ((rs_upad_of(RS_JIT_U) * RS_JIT_R + 15) / 16 + RS_JIT_NT - 1) / RS_JIT_NT : 1;
  constexpr int kPG = kPGsrc > kPGimg ? kPGsrc : kPGimg;
  constexpr bool kPrefetch = FIXED && DIRECT && SCHED == 9 && kPG <= 2;
  /* rs_tti_in.cqi_epoch: this call's reports are the previous call's -- the grid comes from the context's image in HBM, as stored */
  const bool grid_from_image = DIRECT && p.image_mode == 2;
  double pre_avg[kPU];
  int pre_sl[kPU];
  uint4 pre_grid[kPG];
  if constexpr (kPrefetch) {
    const uint4* src = grid_from_image ? (const uint4*)p.grid_image : (const uint4*)p.epochs;
    const int n16 = grid_from_image ? (R * (int)p.Upad + 15) >> 4 : (int)(p.grid_stride >> 4);
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
    /* (a drop-in call carries no state but slice_rbs_offset_: the pending grants, the clock, the rand() ring and the cumulative
     * counters belong to the caller's simulator -- the one-TTI kernel neither loads nor stores them; round 6: the store phase was
     * 7 K of the call's 62 K cycles, its averages going BACK over the host link) */
    s_tx[u] = kDirect ? 0 : p.tx_bytes[(size_t)cell * U + u];
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
  for (int i = tid; i < (R + 1) * 16; i += nt) s_tbs[i] = p.tbs_eff[i];
  /* (the transposing refresh of a drop-in call writes bytes into a zeroed grid; an image, like a batch's grids, carries its padding) */
  if (!grid_from_image)
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
  if (tid == 0) m->grid_free = 0;
  if ((SCHED == 9 || SCHED == 10) && tid < 3) m->heap_sorts[tid] = 0;
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
  double t = kDirect ? 0.0 : scal->t;
  double last_update = kDirect ? 0.0 : scal->last_update;
  long long last_sent = kDirect ? 0 : scal->last_sent;
  int reported = kDirect ? 0 : scal->reported;
  int cqi_row = kDirect ? 0 : scal->cqi_row;
  int served_prev = kDirect ? 0 : scal->served_prev;
  long long n_done = kDirect ? 0 : scal->n_done;
  WaveRng rng;
  rng.r = 0;
  rng.f = kDirect ? 0 : scal->rng_f;
  rng.b = kDirect ? 0 : scal->rng_b;
  if (!kDirect && wave == quota_wave && lane < 31) rng.r = scal->rng_r[lane];
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
  if (kHoldSched || kQuad7) {
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
  if (DIRECT) stamp_acc[9] = stamp_prev - stamp_entry; /* (one-TTI kernels: the load phase; batches keep the slot for the greedy scan's diagnostics, rs_interslice.h) */
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
#include "rs_phase_queues.inc"
  bool have_spec = false; /* this TTI's EWMA, metric scan and quotas were prepared during the previous TTI's serial phase */
  bool have_quota = false; /* held winners: this TTI's quotas were worked out by the quota wave during the previous TTI's serial phase */
  bool have_ewma = false;  /* held winners: ... and so were the PF averages and terms (decay by the idle waves, served users by wave 0) */
  int pre_listed = -1;     /* held winners: this wave's list for this TTI was packed during the previous TTI's serial phase (-1: no) */
  bool have_scan = false;  /* NVS (kEarly17): this TTI's winners were found during the previous TTI's serial phase */
  /* The next CQI grid, fetched ahead (batches on epoch grids; the schedulers whose TTI ends with wave 0 alone): when TTI t+1 starts a
   * new epoch, the other waves read its grid from HBM during TTI t's serial phase and write it over the old one as soon as wave 0
   * has read what its link adaptation needs of it (m->grid_free).  The device-resident grids are stored as the LDS image, RBG-major
   * [R][Upad], so a refresh is a straight 16-byte copy either way; with the fetch ahead TTI t+1 starts with its grid in place -- no
   * HBM latency at the top of the TTI.  Streamed-CQI mode (cqi_refresh = 1), 512 cells x 500 UEs x 25 RBGs, same box
   * (tools/experiments/r04/run20.sh, against -DRS_NO_GRID_AHEAD, which keeps the straight copy at the top): GreedyByRow 59.9 against 53.9 M
   * TTIs/s, NVS 148.9 against 124.4, per-flow PF 156.3 against 125.9; MaximizeCell LOSES (26.7 against 28.5 streamed, 31.2 against
   * 33.4 with the grid resident: its kernel is register-bound and the extra live scalars cost more spills than the fetch saves),
   * so it keeps the copy at the top of the TTI (-DRS_GRID_AHEAD_ALL: MaximizeCell too).
   * The code costs the kernels that never use it (same box, grid resident, with / without it compiled in: NVS 218.7 / 232.4 M TTIs/s,
   * GreedyByRow at 64 RBGs 44.4 / 48.0 -- tools/experiments/r04/run30.sh), so only the kernels of a streamed batch carry it: rs_jit.cpp defines
   * RS_JIT_STREAMED when the batch's cqi_refresh is at most 4 (the built-in kernels copy at the top of the TTI). */
#ifndef RS_JIT_STREAMED
#define RS_JIT_STREAMED 0
#endif
#ifdef RS_GRID_AHEAD_ALL
  constexpr bool kGridAhead = FIXED && RS_JIT_STREAMED && !DIRECT && (SCHED == 1 || SCHED == 7 || SCHED == 8 || SCHED == 9 || SCHED == 101 || SCHED == 103);
#else
  constexpr bool kGridAhead = FIXED && RS_JIT_STREAMED && !DIRECT && (SCHED == 1 || SCHED == 7 || SCHED == 8 || SCHED == 101 || SCHED == 103);
#endif
  bool grid_ahead = false; /* this TTI's grid was written during the previous TTI's serial phase */
  /* (batches only; a launch argument says how many cells one dispatch round holds: 0 = no balancing) */
  const int prio_class = (!DIRECT && p.prio_round_cells > 0) ? ((int)blockIdx.x / p.prio_round_cells) & 1 : 0;
  const bool prio_windows = !DIRECT && p.prio_round_cells > 0 && p.prio_sum == nullptr;
  const bool prio_feedback = !DIRECT && p.prio_sum != nullptr;
  if (prio_feedback && tid == 0) m->prio_boost = 0; /* (the barriers of the first TTI lie between this and its first reader) */
#ifndef RS_PRIO_PERIOD
#define RS_PRIO_PERIOD 16 /* TTIs between two looks at the batch's progress */
#endif
  for (int tti = 0; tti < p.n_ttis; ++tti) {
    bool prio_boost = prio_windows && ((((int)(__builtin_amdgcn_s_memrealtime() >> RS_PRIO_WINDOW_LOG2)) ^ prio_class) & 1) != 0;
    if (prio_feedback) {
      /* Feedback (the default): every RS_PRIO_PERIOD TTIs thread 0 adds its cell's TTIs to the launch's sum and reads it back; behind
       * the average = the sum exceeds my own count times the number of cells (the cells dispatched second, which lose every tie to
       * the older cells on their CUs).  The flag is a word in LDS that every wave reads at the top of the TTI. */
      prio_boost = rs_lds_load(&m->prio_boost) != 0;
    }
    RS_STAMP(11);
    auto prb_ptr = [&](int user, int r2) -> const uint8_t* { /* the G PRBs of RBG r2 as `user` reported them */
      if (DIRECT) return p.prb_cqi + ((size_t)user * R + r2) * G;
      if (p.cqi_mode == RS_CQI_EPOCHS) {
        const long long e = epoch < p.n_epochs ? epoch : (long long)p.n_epochs - 1;
        return p.epochs_prb + ((size_t)cell * p.n_epochs + (size_t)e) * p.grid_stride_prb + ((size_t)user * R + r2) * G;
      }
      return p.trace_prb + (((size_t)p.user_trace[(size_t)cell * U + user] * p.n_rows + cqi_row) * R + r2) * G;
    };
#include "rs_phase_p0_p1.inc"
#include "rs_phase_p2.inc"
#include "rs_phase_nvs_sampler.inc"
#include "rs_phase_p3.inc"
#include "rs_phase_p4.inc"
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
     * what the shorter top saves, as round 2 found for its speculation -- so: GreedyByRow, and MaximizeCell from two users per thread on. */
    /* (MaximizeCell re-measured on the lean build, tools/experiments/r04/run44.sh: 34.45 against 34.64 M at one user per thread, 31.9 against 31.0 at two:
     * from two users per thread on the shorter top of the TTI outweighs what the busy waves cost wave 0) */
    const bool ewma_next = quota_next && (SCHED == 8 || (SCHED == 9 && FIXED && RS_JIT_U > RS_JIT_NT));
    /* ... and pack their lists for TTI t+1 (a one-chunk shape: at most 64 items per wave).  Packing them for MaximizeCell too,
     * without the early EWMA, was measured in round 4: 33.34 against 33.38 M TTIs/s -- the top of its TTI does not wait for it. */
    const bool prelist_next = kHoldSched && ewma_next && hold_ok && n_items_rt <= 64 * nwaves;
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
    /* does TTI t+1 start a new epoch whose grid is in HBM?  Then the other waves fetch it beside wave 0 (kGridAhead) */
    bool grid_next = false;
    long long epoch_next = 0;
#ifndef RS_NO_GRID_AHEAD
    if (kGridAhead && p.cqi_mode == RS_CQI_EPOCHS && nwaves >= 2 && tti + 1 < p.n_ttis && epoch_pos + 1 == p.refresh) {
      epoch_next = epoch + 1;
      if (epoch_next == p.n_epochs && p.epoch_wrap) epoch_next = 0;
      grid_next = epoch_next < p.n_epochs; /* (past the last grid: the next TTI's P0 reports it) */
    }
#endif
    if (SCHED != 10 && wave == 0) {
#include "rs_phase_p4_serial.inc"
#include "rs_phase_p5.inc"
    }
#include "rs_phase_next.inc"
    if (prio_feedback && tid == (nt > 64 ? 64 : 0) && ((tti + 1) & (RS_PRIO_PERIOD - 1)) == 0) {
      /* (the first thread of wave 1, which is idle or early at the end of the TTI: the atomic's round trip is not on wave 0's path) */
      const unsigned long long sum = atomicAdd(p.prio_sum, (unsigned long long)RS_PRIO_PERIOD) + RS_PRIO_PERIOD;
      m->prio_boost = (unsigned long long)(tti + 1) * (unsigned long long)p.n_cells < sum ? 1 : 0;
    }
    RS_SETPRIO(0);
    RS_STAMP(7);
    __syncthreads();
    RS_STAMP(8);
    grid_ahead = grid_next;
    have_spec = kSpecSched && spec_next;
    have_quota = quota_next || (kEarly17 && SCHED == 7 && early17);
    have_ewma = (kHoldSched && ewma_next) || early17;
    if (kEarly17 && SCHED == 7) {
      /* did the scanning waves find TTI t+1's winners?  The same test they made, on values that are final behind the barrier */
      have_scan = early17 && early_scan_ok && rs_lds_load(nvs_word_nxt) != seg_this;
    }
    /* (the packing condition is the same on every wave: wave 0 learns here that wave 1 packed its list) */
    if (wave == 0 && prelist_next) {
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
    if constexpr (kQCumRegs) { /* the launch's cumulative counts of this thread's one user: plain adds, the owner is the only writer */
      if (tid < U) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const size_t bi = bearer_index(tid, b);
          if (qc_bytes[b] != 0) p.b_cumb[bi] += qc_bytes[b];
          if (qc_rbs[b] != 0) p.b_cumr[bi] += qc_rbs[b];
        }
        if (qc_ubytes != 0) p.cum_bytes[(size_t)cell * U + tid] += qc_ubytes;
        if (qc_urbs != 0) p.cum_rbs[(size_t)cell * U + tid] += qc_urbs;
      }
    }
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
  for (int u = tid; u < U && !kCumRegs && !kDirect; u += nt) {
    p.avg[(size_t)cell * U + u] = ((kSpecSched || kHoldSched) && s_avg[u] < 1) ? 1.0 : s_avg[u];
    p.tx_bytes[(size_t)cell * U + u] = s_tx[u];
  }
  if (tid < S) p.slice_state[(size_t)cell * S + tid] = s_sstate[tid];
  if (!kDirect && wave == quota_wave && lane < 31) scal->rng_r[lane] = rng.r;
  if (tid == 0) {
    if (!DIRECT) {
      scal->t = t;
      scal->last_update = last_update;
      scal->last_sent = last_sent;
      scal->reported = reported;
      scal->cqi_row = cqi_row;
      scal->served_prev = served_prev;
      scal->n_done = n_done;
      scal->clk_end = __builtin_readcyclecounter();
      scal->real_end = __builtin_amdgcn_s_memrealtime();
    }
    if (local_err) atomicExch(p.err, local_err);
    if constexpr (SCHED == 9 || SCHED == 10) { /* diagnostics: the sort emulation's heap-sort fallbacks (rs_sort_device.h), normally none */
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (m->heap_sorts[k] != 0) scal->heap_sorts[k] += m->heap_sorts[k];
    }
#ifdef RS_STAMPS
    if (p.stamps) {
      if (DIRECT) stamp_acc[10] += __builtin_readcyclecounter() - stamp_prev; /* (one-TTI kernels: everything behind the TTI's closing barrier = the store phase) */
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
  if (!kDirect && wave == quota_wave && lane == 0) {
    scal->rng_f = rng.f;
    scal->rng_b = rng.b;
  }
  if constexpr (DIRECT) {
    /* rs_schedule_tti's completion word: every thread's outputs are out (system scope) before thread 0 publishes the sequence number */
    if (p.done_flag) {
      __threadfence_system();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.done_flag, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
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
#if defined(RS_JIT_LEAN) && RS_JIT_LEAN && !RS_JIT_DIRECT
  /* The lean build of a batch kernel: the launch's run-time options that the long runs never use are constants here -- epoch grids
   * (no trace rows, no per-PRB twins), no per-TTI decision log, no error-model draws, no synthetic-experiment transport blocks --
   * so their branches, pointers and live scalars are gone (the kernels are bound by registers and issue slots, not by HBM).  The
   * host launches it only when the launch block says exactly this (rs_api.cpp). */
  p.cqi_mode = RS_CQI_EPOCHS;
  p.trace = nullptr; p.trace_prb = nullptr; p.epochs_prb = nullptr; p.user_trace = nullptr;
  p.log_map = nullptr; p.log_quota = nullptr; p.log_target = nullptr; p.log_tbs = nullptr; p.log_uinfo = nullptr; p.log_keys = nullptr;
  p.phy_draws = 0; p.synthetic = 0;
#elif defined(RS_JIT_LEAN) && RS_JIT_LEAN
  /* ... and of a drop-in context's one-TTI kernel: the plain call -- per-RBG reports, no customised slices, no m_requiredRBs /
   * dataToTransmit gates, exponents in {0, 1}, every input an ordinary FP32 number, no UpperBound lists, no synthetic-experiment
   * blocks (rs_schedule_tti picks it per call) */
  p.cqi_mode = RS_CQI_EPOCHS;
  p.prb_cqi = nullptr; p.queue_mode = 0; p.alpha = nullptr; p.beta = nullptr; p.hol = nullptr; p.prio = nullptr;
  p.gate = nullptr; p.exact_scan = 0; p.gen_exp = 0; p.gen_num = nullptr; p.log_upper = nullptr; p.synthetic = 0;
  p.trace = nullptr; p.trace_prb = nullptr; p.epochs_prb = nullptr; p.log_keys = nullptr;
#endif
  constexpr RsCarve kCv = rs_carve(RS_JIT_S, RS_JIT_U, RS_JIT_R, RS_JIT_SCHED, RS_JIT_NT, RS_JIT_CARVEQ, RS_JIT_WIN);
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
                                    int R, int Upad, uint64_t seed, int64_t first_cell, RsCdf cdf) {
  /* the grids are stored as the cell kernel's LDS image: RBG-major [R][Upad], zeros in the padding (rs_upad_of) */
  const int64_t grids = (int64_t)n_cells * n_epochs;
  const int64_t per_grid16 = grid_stride >> 4;
  const int64_t total = grids * per_grid16;
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
    int64_t g = w / per_grid16;
    int64_t o = (w - g * per_grid16) << 4;
    uint8_t out[16];
    int r = (int)(o / Upad), u = (int)(o - (int64_t)r * Upad);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint8_t v = 0;
      if (r < R && u < U) {
        /* keyed by (seed, GLOBAL cell id, epoch, user, rbg): a cell's grids do not depend on how the cells are sharded over
         * ranks nor on how many epochs were generated -- nor on the storage order (the key is the [U][R] index) */
        const int64_t idx = (int64_t)u * R + r;
        const uint64_t gcell = (uint64_t)(first_cell + g / n_epochs), ep = (uint64_t)(g % n_epochs);
        uint64_t h = splitmix64(seed ^ splitmix64(splitmix64(gcell * 0x100000001B3ull + ep) + (uint64_t)idx));
        uint32_t x = (uint32_t)(h >> 32);
        int q = 0;
#pragma unroll
        for (int j = 0; j < 14; ++j) q += x > cdf.c[j] ? 1 : 0;
        v = (uint8_t)(q + 1);
      }
      out[k] = v;
      if (++u == Upad) { u = 0; ++r; }
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

extern "C" hipError_t rs_launch_synth(uint8_t* epochs, int64_t grid_stride, int n_cells, int n_epochs, int U, int R, int Upad,
                                      uint64_t seed, int64_t first_cell, const uint32_t* cdf16, hipStream_t stream) {
  RsCdf cdf;
  for (int i = 0; i < 16; ++i) cdf.c[i] = cdf16[i];
  int64_t total = (int64_t)n_cells * n_epochs * (grid_stride >> 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(rs_synth_cqi_kernel, dim3(blocks), dim3(256), 0, stream, epochs, grid_stride, n_cells, n_epochs,
                     U, R, Upad, seed, first_cell, cdf);
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
