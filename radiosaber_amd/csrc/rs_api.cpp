/*
 * rs_api.cpp -- host side of the C ABI declared in include/radiosaber_hip.h.
 *
 * Validates configurations, owns the device buffers (cell state, CQI grids, traces), evaluates the
 * link-adaptation tables with the host libm (the only place transcendental functions run), seeds the
 * per-cell libc-compatible rand() streams and launches the gfx950 kernels of rs_kernels.hip.
 * There is no CPU implementation of the scheduling path here: without a HIP device every compute
 * entry point fails with RS_ERR_NO_DEVICE.
 *
 * Build with -ffp-contract=off (the table arithmetic must round like the reference's -O0 SSE2 code).
 */
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <fstream>
#include <vector>

#include "../../include/radiosaber_hip.h"
#include "rs_amc_tables.inc"
#include "rs_link_pinned.inc"
#include "rs_device.h"

extern "C" hipError_t rs_launch_cells(const RsLaunch* p, int threads, hipStream_t stream);
extern "C" hipError_t rs_prepare_kernels(int max_lds_bytes);
struct RsJitKernel;
extern "C" RsJitKernel* rs_jit_get(int device, int S, int U, int R, int G, int NT, int sched, int qmode, int win, char* err, size_t errlen,
                                   int flags, /* bit 0: drop-in (one-TTI) kernel, bit 1: streamed batch (cqi_refresh <= 4), bit 2: lean build */
                                   const char* variant = nullptr); /* an autotune candidate: extra -D options and / or "ss=<LLVM scheduler strategy>" */
extern "C" int rs_jit_is_untuned(const RsJitKernel* k);
extern "C" int rs_jit_is_verified(const RsJitKernel* k);  /* carries the self-check mark (this process, or its cache file) */
extern "C" int rs_jit_was_loaded(const RsJitKernel* k);   /* came from the disk cache */
extern "C" void rs_jit_mark_verified(RsJitKernel* k);
extern "C" void rs_jit_reject(RsJitKernel* k);
extern "C" hipError_t rs_jit_launch(RsJitKernel* k, const RsLaunch* p, hipStream_t stream);
extern "C" hipError_t rs_launch_synth(uint8_t* epochs, int64_t grid_stride, int n_cells, int n_epochs, int U, int R, int Upad,
                                      uint64_t seed, int64_t first_cell, const uint32_t* cdf16, hipStream_t stream);
extern "C" hipError_t rs_launch_copy_probe(const void* src, void* dst, size_t bytes, hipStream_t stream);
extern "C" hipError_t rs_launch_slice_bytes(const int64_t* cum_bytes, const uint8_t* user_slice, int n_cells, int U,
                                            int S, unsigned long long* d_out, hipStream_t stream);

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e__ = (expr);                                                                   \
    if (e__ != hipSuccess) return fail(RS_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e__));   \
  } while (0)

/* scratch device buffers / events of one call: released on every return path */
struct ScratchBuffers {
  std::vector<void*> ptrs;
  ~ScratchBuffers() { for (void* q : ptrs) (void)hipFree(q); }
  template <typename T> hipError_t alloc(T** out, size_t bytes) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
    if (e == hipSuccess) { ptrs.push_back(q); *out = (T*)q; }
    return e;
  }
};
struct ScratchEvents {
  std::vector<hipEvent_t> ev;
  ~ScratchEvents() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
  hipError_t create(int n) {
    for (int i = 0; i < n; i++) {
      hipEvent_t e = nullptr;
      hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      ev.push_back(e);
    }
    return hipSuccess;
  }
};

const int kCqiToMcs[15] = {RS_AMC_CQI_TO_MCS};
const double kSinrForCqi[15] = {RS_AMC_SINR_FOR_CQI};
const int kMcsToItbs[29] = {RS_AMC_MCS_TO_ITBS};
const int kTbs[110][27] = {RS_AMC_TBS_TABLE};

/* the reference's effective-SINR expression on x = mean(exp(-sinr_lin))
 * (src/utility/eesm-effective-sinr.h:43-44 with beta = 1) */
double eesm_db_of_mean(double x) {
  double beta = 1;
  double eff = -beta * log(x);
  eff = 10 * log10(eff);
  return eff;
}

double from_bits(uint64_t b) {
  double d;
  memcpy(&d, &b, 8);
  return d;
}
uint64_t to_bits(double d) {
  uint64_t b;
  memcpy(&b, &d, 8);
  return b;
}

/* glibc TYPE_3 rand() (glibc 2.35 stdlib/random_r.c), host copy used to seed the device rings */
struct HostRng {
  uint32_t r[31];
  int f, b;
  void seed(unsigned s) {
    if (s == 0) s = 1;
    int32_t w = (int32_t)s;
    r[0] = (uint32_t)w;
    for (int i = 1; i < 31; i++) {
      long hi = w / 127773, lo = w % 127773;
      long word = 16807 * lo - 2836 * hi;
      if (word < 0) word += 2147483647;
      w = (int32_t)word;
      r[i] = (uint32_t)w;
    }
    f = 3;
    b = 0;
    for (int i = 0; i < 310; i++) next();
  }
  int next() {
    uint32_t v = r[f] += r[b];
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    return (int)(v >> 1);
  }
};

int round_up(int x, int a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

const char* rs_last_error(void) { return g_err; }
int rs_abi_version(void) { return RS_ABI_VERSION; }

int rs_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* ref: src/utility/eesm-effective-sinr.h:82-103 (the reference throws above 512 PRBs) */
int rs_get_rbg_size(int nb_rbs) {
  if (nb_rbs < 1) return fail(RS_ERR_INVALID, "nb_rbs %d < 1", nb_rbs);
  if (nb_rbs <= 10) return 1;
  if (nb_rbs <= 26) return 2;
  if (nb_rbs <= 63) return 3;
  if (nb_rbs <= 110) return 4;
  if (nb_rbs <= 512) return 8;
  return fail(RS_ERR_INVALID, "nb_rbs %d > 512: the reference's get_rbg_size throws", nb_rbs);
}

/* ref: src/core/spectrum/bandwidth-manager.cpp:30-38, 52-108: PRBs of a downlink bandwidth; any other value falls back to 5 MHz */
int rs_dl_prbs_for_bandwidth(double bw_mhz) {
  if (bw_mhz == 1.4) return 6;
  if (bw_mhz == 3) return 15;
  if (bw_mhz == 5) return 25;
  if (bw_mhz == 10) return 50;
  if (bw_mhz == 15) return 75;
  if (bw_mhz == 20) return 100;
  if (bw_mhz == 100) return 512;
  return 25;
}

/* the EESM constants as glibc 2.35 (x86-64) evaluates them: E[1..15], X[1..13] (radiosaber_amd/data/link_tables_glibc_2_35.json ->
 * rs_link_pinned.inc by build.py; the same hex values as SURVEY.md Appendix A / tests/golden/appendix_a.json) */
static const double kPinnedE[15] = {RS_LINK_PINNED_E};
static const double kPinnedX[13] = {RS_LINK_PINNED_X};

int rs_link_tables_pinned(double eff[16], double kbps[16], double eesm_e[16], double eesm_x[16]) {
  eff[0] = kbps[0] = eesm_e[0] = eesm_x[0] = 0;
  for (int c = 1; c <= 15; c++) {
    const int bits = kTbs[0][kMcsToItbs[kCqiToMcs[c - 1]]];
    const double e = (bits / 0.001) / 180000.; /* IEEE divisions of table integers: no libm */
    eff[c] = e;
    kbps[c] = e * 180000 / 1000;
    eesm_e[c] = kPinnedE[c - 1];
  }
  eesm_x[14] = eesm_x[15] = 0;
  for (int k = 1; k <= 13; k++) eesm_x[k] = kPinnedX[k - 1];
  return RS_OK;
}

int rs_link_tables_compare(char* msg, size_t msglen) {
  double eff[16], kbps[16], e[16], x[16];
  if (msg && msglen) msg[0] = 0;
  const int rc = rs_link_tables(eff, kbps, e, x);
  if (rc) return rc;
  int diff = 0;
  std::string text;
  char line[160];
  for (int c = 1; c <= 15; c++)
    if (to_bits(e[c]) != to_bits(kPinnedE[c - 1])) {
      diff++;
      snprintf(line, sizeof line, "E[%d] host %a pinned %a; ", c, e[c], kPinnedE[c - 1]);
      text += line;
    }
  for (int k = 1; k <= 13; k++)
    if (to_bits(x[k]) != to_bits(kPinnedX[k - 1])) {
      diff++;
      snprintf(line, sizeof line, "X[%d] host %a pinned %a; ", k, x[k], kPinnedX[k - 1]);
      text += line;
    }
  if (msg && msglen) snprintf(msg, msglen, "%s", text.c_str());
  return diff;
}

int rs_link_tables(double eff[16], double kbps[16], double eesm_e[16], double eesm_x[16]) {
  eff[0] = kbps[0] = eesm_e[0] = eesm_x[0] = 0;
  for (int c = 1; c <= 15; c++) {
    int bits = kTbs[0][kMcsToItbs[kCqiToMcs[c - 1]]];
    double e = (bits / 0.001) / 180000.; /* AMCModule.cpp:320-327 */
    eff[c] = e;
    kbps[c] = e * 180000 / 1000; /* downlink-transport-scheduler.cpp:688 */
    double s = pow(10, kSinrForCqi[c - 1] / 10);
    double beta = 1;
    eesm_e[c] = exp(-s / beta); /* eesm-effective-sinr.h:38-40 */
  }
  eesm_x[14] = eesm_x[15] = 0;
  for (int k = 1; k <= 13; k++) {
    /* largest positive double x with eesm_db_of_mean(x) >= SINR[k] (the function decreases in x) */
    const double thr = kSinrForCqi[k];
    uint64_t lo = 1, hi = to_bits(1.0) - 1; /* smallest subnormal .. largest double below 1 */
    if (!(eesm_db_of_mean(from_bits(lo)) >= thr)) return fail(RS_ERR_INVALID, "EESM threshold %d unreachable", k);
    while (hi - lo > 1) {
      uint64_t mid = lo + (hi - lo) / 2;
      if (eesm_db_of_mean(from_bits(mid)) >= thr) lo = mid; else hi = mid;
    }
    if (eesm_db_of_mean(from_bits(hi)) >= thr) lo = hi;
    /* the comparison on x is only equivalent to the reference's comparison in dB if the host
     * libm composite is monotone around the threshold: check a +-4096 ulp neighbourhood */
    for (int d = 1; d <= 4096; d++) {
      if (!(eesm_db_of_mean(from_bits(lo - d)) >= thr) || (eesm_db_of_mean(from_bits(lo + d)) >= thr))
        return fail(RS_ERR_INVALID, "host libm not monotone around EESM threshold %d (offset %d ulp)", k, d);
    }
    eesm_x[k] = from_bits(lo);
  }
  for (int k = 2; k <= 13; k++)
    if (!(eesm_x[k] < eesm_x[k - 1])) return fail(RS_ERR_INVALID, "EESM thresholds not decreasing at %d", k);
  return RS_OK;
}

}  // extern "C"

/* ===================================================================================== */

struct rs_batch {
  rs_batch_config cfg;
  std::vector<double> weight;
  std::vector<int32_t> eps, psi, u2s, alpha, beta;
  bool any_alpha = false;
  bool synthetic = false;    /* rs_config.synthetic_exp */
  bool gen_exp = false;      /* drop-in context with an exponent outside {0, 1} (ref: packet-scheduler.h:38-49 takes any int) */
  double* d_gen_num = nullptr; /* [S][16] pow(kbps[cqi], algo_epsilon[s]), host libm */
  int32_t *d_alpha = nullptr, *d_beta = nullptr;
  int S, U, R, G, sched, n_cells, threads;
  bool direct = false;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  /* device */
  RsTables* d_tab = nullptr;
  double* d_weight = nullptr;
  int32_t *d_eps = nullptr, *d_psi = nullptr;
  uint8_t* d_user_slice = nullptr;
  int32_t* d_tbs_eff = nullptr;
  double* d_avg = nullptr;
  int32_t* d_tx = nullptr;
  int64_t *d_cumb = nullptr, *d_cumr = nullptr;
  double* d_sstate = nullptr;
  RsCellScalars* d_scal = nullptr;
  uint8_t* d_epochs = nullptr;
  int64_t grid_stride = 0;
  int32_t n_epochs = 0;
  uint8_t* d_trace = nullptr;
  int32_t n_traces = 0, n_rows = 0, row_mod = 0;
  int32_t* d_user_trace = nullptr;
  uint8_t *d_epochs_prb = nullptr, *d_trace_prb = nullptr; /* per-PRB twins (link adaptation) */
  int64_t grid_stride_prb = 0;
  int32_t cqi_mode = RS_CQI_NONE;
  int32_t* d_err = nullptr;
  unsigned long long* d_slice_bytes = nullptr;
  unsigned long long* d_stamps = nullptr;
  /* finite queues (rs_batch_set_bearers / rs_batch_set_arrivals) */
  bool queues = false;
  int qmode = 2;             /* rs_carve's `queue` argument once the queue model is on: 2 = bearers' hot words in LDS when they fit, 3 = in HBM */
  uint8_t* d_bearer_kind = nullptr;
  int64_t* d_arr_off = nullptr;
  double* d_arr_time = nullptr;
  int32_t *d_arr_nfull = nullptr, *d_arr_last = nullptr;
  int32_t* d_qi = nullptr;   /* 7 arrays [cells][2][U]: head, tail, pk, frag, bytes, pkts, tx */
  double* d_bavg = nullptr;  /* [cells][2][U] */
  int64_t* d_bcum = nullptr; /* 2 arrays [cells][2][U]: bytes, rbs */
  uint8_t* d_qflags = nullptr;
  double* d_qhol = nullptr;
  RsJitKernel* jit = nullptr; /* shape-specialised kernel (owned by the process-wide cache) */
  RsJitKernel* jit_lean = nullptr; /* its lean build, compiled at the first launch that can use it (launch()) */
  bool jit_lean_tried = false;
  bool autotuned = false;       /* rs_batch_config.autotune: the candidates were timed (or the tuning does not apply) */
  bool selfchecked = false;     /* the self-check of the run-time builds: done (or it does not apply) */
  bool selfchecked_general = false, selfchecked_lean = false;
  bool in_selfcheck = false;    /* its trial launches come back through launch(): not again */
  bool jit_rejected = false;    /* ... and a run-time build disagreed with the built-in kernels: dropped */
  char selfcheck_msg[160] = "";
  int autotune_n = 0;
  char autotune_msg[768] = "";
  unsigned long long* d_prio_sum = nullptr; /* TTIs done by all cells of the running launch (the cells' issue-priority feedback) */
  bool jit_wanted = false;
  char jit_msg[512] = ""; /* why the shape-specialised kernel is not in use (empty: it is, or it was not asked for) */
  int64_t ttis_done = 0;
  RsLaunch base{};
};

namespace {

int validate(const rs_config* c, bool direct) {
  if (!c) return fail(RS_ERR_INVALID, "null config");
  if (c->n_slices < 1 || c->n_slices > RS_MAX_SLICES) return fail(RS_ERR_INVALID, "n_slices %d outside 1..%d", c->n_slices, RS_MAX_SLICES);
  if (c->n_users < 1 || c->n_users > RS_MAX_USERS) return fail(RS_ERR_INVALID, "n_users %d outside 1..%d", c->n_users, RS_MAX_USERS);
  if (c->n_rbgs < 1 || c->n_rbgs > RS_MAX_RBGS) return fail(RS_ERR_INVALID, "n_rbgs %d outside 1..%d", c->n_rbgs, RS_MAX_RBGS);
  if (c->rbg_size < 1 || c->rbg_size > 8) return fail(RS_ERR_INVALID, "rbg_size %d outside 1..8", c->rbg_size);
  if (c->n_rbgs * c->rbg_size > 512) return fail(RS_ERR_INVALID, "more than 512 PRBs (reference get_rbg_size throws)");
  if ((c->synthetic_exp | 1) != 1) return fail(RS_ERR_INVALID, "synthetic_exp %d is not 0 or 1 (an rs_config of an older ABI? it gained the field in ABI 8)", c->synthetic_exp);
  if (c->link_tables < RS_LINK_DEFAULT || c->link_tables > RS_LINK_PINNED_GLIBC_2_35)
    return fail(RS_ERR_INVALID, "link_tables %d is not an RS_LINK_* value (an rs_config of an older ABI? it gained the field in ABI 11)", c->link_tables);
  if (c->sched != RS_SCHED_PF && c->sched != RS_SCHED_NVS && c->sched != RS_SCHED_SEQUENTIAL && c->sched != RS_SCHED_MAXCELL &&
      c->sched != RS_SCHED_VOGEL && c->sched != RS_SCHED_SUBOPT && c->sched != RS_SCHED_UPPERBOUND && c->sched != RS_SCHED_NVS_NONGREEDY)
    return fail(RS_ERR_INVALID, "sched %d not supported (1, 7, 8, 9, 10, 11, 101, 103)", c->sched);
  if (!c->slice_weight || !c->algo_alpha || !c->algo_epsilon || !c->algo_psi || !c->user_to_slice)
    return fail(RS_ERR_INVALID, "null slice/user array");
  for (int s = 0; s < c->n_slices; s++) {
    if ((c->algo_alpha[s] | 1) != 1 || (c->algo_alpha[s] && (!c->algo_beta || (c->algo_beta[s] | 1) != 1)))
      return fail(RS_ERR_INVALID, "slice %d: algo_alpha/algo_beta must be 0 or 1", s);
    /* batches update the PF averages on the device, which has no pow(): exponents 0 and 1 (exact in libm) only; a drop-in context
     * takes any integers -- the host's libm raises the 16 numerators per slice once and every user's denominator per call */
    if (!direct && ((c->algo_epsilon[s] | 1) != 1 || (c->algo_psi[s] | 1) != 1))
      return fail(RS_ERR_INVALID, "slice %d: algo_epsilon/algo_psi must be 0 or 1 in a batch (any integer in a drop-in context)", s);
    if (c->algo_epsilon[s] < -64 || c->algo_epsilon[s] > 64 || c->algo_psi[s] < -64 || c->algo_psi[s] > 64)
      return fail(RS_ERR_INVALID, "slice %d: algo_epsilon/algo_psi outside -64..64", s);
  }
  for (int u = 0; u < c->n_users; u++) {
    int s = c->user_to_slice[u];
    if (s < 0 || s >= c->n_slices) return fail(RS_ERR_INVALID, "user %d: slice %d out of range", u, s);
    if (u && s < c->user_to_slice[u - 1]) return fail(RS_ERR_INVALID, "user_to_slice must be non-decreasing (user %d)", u);
  }
  return RS_OK;
}

int upad_of(int U) { return rs_upad_of(U); }

/* the longest slice window of the batch in the stage-1 reciprocal array: 8-aligned start of the slice's first user to the
 * 8-aligned end behind its last (a compile-time constant of the shape-specialised kernel, RS_JIT_WIN) */
int slice_window(const rs_batch* b) {
  int win = 0;
  for (int u = 0; u < b->U;) {
    int e = u;
    while (e < b->U && b->u2s[e] == b->u2s[u]) e++;
    const int w = ((e + 7) & ~7) - (u & ~7);
    win = w > win ? w : win;
    u = e;
  }
  return win;
}

void carve_lds(rs_batch* b, RsLaunch* L) {
  /* (drop-in contexts of schedulers 1 and 7 carry the gate scratch too: rs_tti_in.required_rbs / data_to_transmit) */
  const bool gate_scratch = b->direct && (b->sched == RS_SCHED_PF || b->sched == RS_SCHED_NVS);
  /* the same arguments the batch's shape-specialised kernels are compiled with (rs_jit_get: slice_window() for batches, 0 for
   * drop-in contexts); a batch that runs the built-in kernels splits NVS slices earlier (RS_NVS_WHOLE_SLICE_BUILTIN) */
  const RsCarve c = rs_carve(b->S, b->U, b->R, b->sched, b->threads, b->queues ? b->qmode : (gate_scratch ? 1 : 0),
                             b->direct ? 0 : slice_window(b), (b->jit_wanted || b->direct) ? RS_NVS_WHOLE_SLICE : RS_NVS_WHOLE_SLICE_BUILTIN);
  L->Upad = c.Upad;
  L->nvs_seg = c.nvs_seg;
  L->off_avgk = c.off_avgk; L->off_rcp = c.off_rcp; L->off_tab = c.off_tab; L->off_slice = c.off_slice;
  L->off_tx = c.off_tx; L->off_misc = c.off_misc; L->off_tbs = c.off_tbs; L->off_elems = c.off_elems;
  L->off_sorted = c.off_sorted; L->off_items = c.off_items; L->off_sortx = c.off_sortx; L->off_cqi = c.off_cqi;
  L->off_queue = c.off_queue;
  L->off_qstate = c.off_qstate;
  L->q_lds = c.q_lds;
  L->lds_bytes = c.lds_bytes;
  L->n_seg = c.n_seg;
  L->n_items = c.n_items;
}

int init_scalars(rs_batch* b, const uint32_t* seed, const int64_t* skip) {
  std::vector<RsCellScalars> sc(b->n_cells);
  double t = 0;
  for (int k = 0; k < b->cfg.first_tti; k++) t += 0.001; /* simulator.cc:117-126 */
  for (int c = 0; c < b->n_cells; c++) {
    RsCellScalars& s = sc[c];
    memset(&s, 0, sizeof s);
    s.t = t;
    s.last_update = 0.1; /* bearers are created by the application start event at 0.1 s */
    HostRng g;
    g.seed(seed ? seed[c] : 1u);
    if (skip)
      for (int64_t i = 0; i < skip[c]; i++) g.next();
    memcpy(s.rng_r, g.r, sizeof g.r);
    s.rng_f = g.f;
    s.rng_b = g.b;
  }
  HIP_TRY(hipMemcpy(b->d_scal, sc.data(), sizeof(RsCellScalars) * b->n_cells, hipMemcpyHostToDevice));
  return RS_OK;
}

int batch_alloc(rs_batch* b) {
  const size_t cells = b->n_cells, U = b->U, S = b->S;
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  if (b->cfg.cell.stream) {
    b->stream = (hipStream_t)b->cfg.cell.stream;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->own_stream = true;
  }
  RsTables t;
  memset(&t, 0, sizeof t);
  double eff[16];
  /* the EESM constants: pinned (glibc 2.35) for batches, this host's libm for a drop-in beside a locally built reference, unless the
   * configuration says otherwise (rs_config.link_tables) */
  const int link = b->cfg.cell.link_tables != RS_LINK_DEFAULT ? b->cfg.cell.link_tables : (b->direct ? RS_LINK_HOST_LIBM : RS_LINK_PINNED_GLIBC_2_35);
  int rc = link == RS_LINK_PINNED_GLIBC_2_35 ? rs_link_tables_pinned(eff, t.kbps, t.eesm_e, t.eesm_x) : rs_link_tables(eff, t.kbps, t.eesm_e, t.eesm_x);
  if (rc) return rc;
  for (int c = 1; c <= 15; c++) {
    t.pfnum[c] = eff[c] * 180000.; /* dl-pf-packet-scheduler.cpp:138 */
    t.eff[c] = eff[c];
    t.mcs_of_cqi[c] = kCqiToMcs[c - 1];
    t.itbs_of_cqi[c] = kMcsToItbs[kCqiToMcs[c - 1]];
    t.tbs1_of_cqi[c] = kTbs[0][kMcsToItbs[kCqiToMcs[c - 1]]];
    /* ref: downlink-transport-scheduler.cpp:643-658 with AMCModule.cpp:253-268: the PRB's SINR is the table value of its CQI, and
     * the CQI read back from it counts the thresholds at or below it */
    int rt = 1;
    while (rt <= 14 && kSinrForCqi[rt] <= kSinrForCqi[c - 1]) rt++;
    t.tbs1_syn[c] = kTbs[0][kMcsToItbs[kCqiToMcs[rt - 1]]];
  }
  /* AMCModule.cpp:313-315 reads TransportBlockSizeTable[-1][itbs] when nbRBs > 110 and nbRBs % 5 == 0.
   * In the as-shipped -O0 build McsToItbs[29] sits 128 bytes in front of the table, so
   * T[-1][i] = McsToItbs[5+i] for i <= 23 and 0 (padding) beyond. */
  for (int i = 0; i < 27; i++) t.tbs_row_m1[i] = i <= 23 ? kMcsToItbs[5 + i] : 0;
  HIP_TRY(hipMalloc(&b->d_tab, sizeof t));
  HIP_TRY(hipMemcpy(b->d_tab, &t, sizeof t, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_weight, 8 * S));
  HIP_TRY(hipMemcpy(b->d_weight, b->weight.data(), 8 * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_eps, 4 * S));
  HIP_TRY(hipMemcpy(b->d_eps, b->eps.data(), 4 * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_psi, 4 * S));
  HIP_TRY(hipMemcpy(b->d_psi, b->psi.data(), 4 * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_alpha, 4 * S));
  HIP_TRY(hipMemcpy(b->d_alpha, b->alpha.data(), 4 * S, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_beta, 4 * S));
  HIP_TRY(hipMemcpy(b->d_beta, b->beta.data(), 4 * S, hipMemcpyHostToDevice));
  std::vector<uint8_t> us(U);
  for (size_t u = 0; u < U; u++) us[u] = (uint8_t)b->u2s[u];
  HIP_TRY(hipMalloc(&b->d_user_slice, U));
  HIP_TRY(hipMemcpy(b->d_user_slice, us.data(), U, hipMemcpyHostToDevice));
  {
    /* TBS bits of n RBGs = n*G PRBs per itbs (AMCModule.cpp:306-317); row 0 unused.  The device gets them with the CQI -> MCS -> I_TBS
     * step folded in, [R+1][16] indexed by final CQI: the kernel's LDS table is then a straight copy (round 6: the two dependent loads
     * per entry -- itbs_of_cqi, then the row -- were ~1 us of every one-TTI launch) */
    std::vector<int32_t> te((size_t)(b->R + 1) * 27, 0);
    for (int n = 1; n <= b->R; n++)
      for (int i = 0; i < 27; i++) {
        const int nprb = n * b->G;
        int v;
        if (nprb <= 110) v = kTbs[nprb - 1][i];
        else v = 5 * kTbs[nprb / 5 - 1][i] + (nprb % 5 == 0 ? t.tbs_row_m1[i] : kTbs[nprb % 5 - 1][i]);
        te[(size_t)n * 27 + i] = v;
      }
    for (int32_t v : te)
      if (v / 8 > RS_MAX_BYTES_PER_TTI) return fail(RS_ERR_INVALID, "a transport block of %d bits does not fit the per-launch byte counters", v);
    std::vector<int32_t> te16((size_t)(b->R + 1) * 16, 0);
    for (int n = 0; n <= b->R; n++)
      for (int c = 0; c < 16; c++) te16[(size_t)n * 16 + c] = te[(size_t)n * 27 + t.itbs_of_cqi[c]];
    HIP_TRY(hipMalloc(&b->d_tbs_eff, 4 * te16.size()));
    HIP_TRY(hipMemcpy(b->d_tbs_eff, te16.data(), 4 * te16.size(), hipMemcpyHostToDevice));
  }
  if (b->gen_exp) {
    /* ref: downlink-transport-scheduler.cpp:688-693  pow(spectralEfficiency * 180000 / 1000, epsilon), with this host's libm --
     * the library the reference itself calls */
    std::vector<double> gn((size_t)S * 16, 0.0);
    for (size_t s = 0; s < S; s++)
      for (int c = 1; c <= 15; c++) gn[s * 16 + c] = pow(t.kbps[c], b->eps[s]);
    HIP_TRY(hipMalloc(&b->d_gen_num, 8 * gn.size()));
    HIP_TRY(hipMemcpy(b->d_gen_num, gn.data(), 8 * gn.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMalloc(&b->d_avg, 8 * cells * U));
  HIP_TRY(hipMalloc(&b->d_tx, 4 * cells * U));
  HIP_TRY(hipMalloc(&b->d_cumb, 8 * cells * U));
  HIP_TRY(hipMalloc(&b->d_cumr, 8 * cells * U));
  HIP_TRY(hipMalloc(&b->d_sstate, 8 * cells * S));
  HIP_TRY(hipMalloc(&b->d_scal, sizeof(RsCellScalars) * cells));
  HIP_TRY(hipMalloc(&b->d_err, 4));
  HIP_TRY(hipMalloc(&b->d_slice_bytes, 8 * 64));
  HIP_TRY(hipMemset(b->d_err, 0, 4));
#ifdef RS_STAMPS
  HIP_TRY(hipMalloc(&b->d_stamps, 8 * 20 * (size_t)b->n_cells));
  HIP_TRY(hipMemset(b->d_stamps, 0, 8 * 20 * (size_t)b->n_cells));
#endif
  std::vector<double> avg(cells * U, 100000.0); /* radio-bearer.cpp:54 */
  HIP_TRY(hipMemcpy(b->d_avg, avg.data(), 8 * cells * U, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(b->d_tx, 0, 4 * cells * U));
  HIP_TRY(hipMemset(b->d_cumb, 0, 8 * cells * U));
  HIP_TRY(hipMemset(b->d_cumr, 0, 8 * cells * U));
  HIP_TRY(hipMemset(b->d_sstate, 0, 8 * cells * S));
  rc = init_scalars(b, nullptr, nullptr);
  if (rc) return rc;

  RsLaunch& L = b->base;
  memset(&L, 0, sizeof L);
  L.S = b->S; L.U = b->U; L.R = b->R; L.G = b->G;
  L.sched = b->sched;
  L.n_cells = b->n_cells;
  L.refresh = b->cfg.cqi_refresh;
  L.epoch_wrap = b->cfg.cqi_epoch_wrap ? 1 : 0;
  L.phy_draws = b->cfg.phy_error_draws;
  L.tab = b->d_tab; L.weight = b->d_weight; L.eps = b->d_eps; L.psi = b->d_psi;
  L.alpha = b->d_alpha; L.beta = b->d_beta;
  L.user_slice = b->d_user_slice;
  L.tbs_eff = b->d_tbs_eff;
  L.synthetic = b->synthetic ? 1 : 0;
  L.avg = b->d_avg; L.tx_bytes = b->d_tx; L.cum_bytes = b->d_cumb; L.cum_rbs = b->d_cumr;
  L.slice_state = b->d_sstate; L.scal = b->d_scal; L.err = b->d_err; L.stamps = b->d_stamps;
  carve_lds(b, &L);
  if (L.lds_bytes > 160 * 1024) return fail(RS_ERR_INVALID, "cell needs %d B of LDS (> 160 KiB)", L.lds_bytes);
  {
    /* co-resident cells take turns at the issue priority (RS_SETPRIO, rs_kernels.hip) when the batch has more cells than the device has
     * compute units -- the second dispatch round shares its CUs with the first, which would win every tie by age.  RS_PRIO_BALANCE=0 / 1
     * overrides. */
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, b->cfg.cell.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    int mode = (!b->direct && b->n_cells > cus) ? 2 : 0; /* 2: feedback through a progress sum (rs_kernels.hip), 1: time windows, 0: off */
    if (const char* e = getenv("RS_PRIO_BALANCE")) mode = b->direct ? 0 : atoi(e);
    L.prio_round_cells = mode == 1 ? cus : 0;
    if (mode == 2) {
      HIP_TRY(hipMalloc(&b->d_prio_sum, 8));
      HIP_TRY(hipMemset(b->d_prio_sum, 0, 8));
      L.prio_sum = b->d_prio_sum;
    }
  }
  HIP_TRY(rs_prepare_kernels(160 * 1024));
  return RS_OK;
}

rs_batch* batch_new(const rs_batch_config* cfg, bool direct) {
  if (!cfg) { fail(RS_ERR_INVALID, "null config"); return nullptr; }
  if (validate(&cfg->cell, direct)) return nullptr;
  if (cfg->n_cells < 1) { fail(RS_ERR_INVALID, "n_cells %d < 1", cfg->n_cells); return nullptr; }
  if (!direct && cfg->cqi_refresh < 1) { fail(RS_ERR_INVALID, "cqi_refresh %d < 1", cfg->cqi_refresh); return nullptr; }
  if (cfg->first_tti < 0) { fail(RS_ERR_INVALID, "first_tti %d < 0", cfg->first_tti); return nullptr; }
  int threads = cfg->threads_per_cell;
  if (threads == 0) {
    /* default: 512 threads per cell while the batch leaves at most ~2 cells per CU (per-level latency
     * matters), 256 (two array positions per lane, fewer instructions per level) once 4+ cells share a CU */
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, cfg->cell.device) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount;
    threads = (!direct && cfg->n_cells >= 4 * cus) ? 256 : 512;
  }
  int want_jit = cfg->jit;
  if (const char* e = getenv("RS_JIT")) want_jit = atoi(e);
  /* the kernels built into the library are bounded at 512 threads; a shape-specialised build takes up to 1 024 (the 64-RBG
   * grid runs 640: two sort positions per lane instead of three on half of the waves) */
  const int max_threads = (want_jit && !direct) ? 1024 : 512;
  if (threads % 64 || threads < 64 || threads > max_threads) {
    fail(RS_ERR_INVALID, "threads_per_cell %d (a multiple of 64 in 64..%d%s)", threads, max_threads,
         max_threads == 512 ? "; up to 1024 with jit = 1" : "");
    return nullptr;
  }
  if ((cfg->cqi_epoch_wrap | 1) != 1) { fail(RS_ERR_INVALID, "cqi_epoch_wrap %d is not 0 or 1", cfg->cqi_epoch_wrap); return nullptr; }
  if ((cfg->autotune | 1) != 1) { fail(RS_ERR_INVALID, "autotune %d is not 0 or 1", cfg->autotune); return nullptr; }
  if (cfg->selfcheck < -1 || cfg->selfcheck > 1) { fail(RS_ERR_INVALID, "selfcheck %d outside -1..1", cfg->selfcheck); return nullptr; }
  if (cfg->queue_state_lds < -1 || cfg->queue_state_lds > 1) { fail(RS_ERR_INVALID, "queue_state_lds %d outside -1..1", cfg->queue_state_lds); return nullptr; }
  if (cfg->cell.sched == RS_SCHED_UPPERBOUND) {
    /* the per-slice sorts use the register form of the sort emulation: at most four array positions per thread */
    const int N = cfg->cell.n_rbgs * cfg->cell.n_slices;
    if (cfg->threads_per_cell == 0)
      while (threads < 512 && N > 4 * threads) threads += 64;
    if (N > 4 * threads) {
      fail(RS_ERR_INVALID, "RS_SCHED_UPPERBOUND: n_rbgs * n_slices = %d exceeds 4 * threads_per_cell = %d", N, 4 * threads);
      return nullptr;
    }
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { fail(RS_ERR_NO_DEVICE, "no HIP device"); return nullptr; }
  if (cfg->cell.device < 0 || cfg->cell.device >= n) { fail(RS_ERR_NO_DEVICE, "device %d of %d", cfg->cell.device, n); return nullptr; }
  rs_batch* b = new (std::nothrow) rs_batch();
  if (!b) { fail(RS_ERR_INVALID, "out of memory"); return nullptr; }
  b->cfg = *cfg;
  const rs_config& c = cfg->cell;
  b->S = c.n_slices; b->U = c.n_users; b->R = c.n_rbgs; b->G = c.rbg_size; b->sched = c.sched;
  b->n_cells = cfg->n_cells;
  b->threads = threads;
  b->direct = direct;
  b->weight.assign(c.slice_weight, c.slice_weight + b->S);
  b->eps.assign(c.algo_epsilon, c.algo_epsilon + b->S);
  b->psi.assign(c.algo_psi, c.algo_psi + b->S);
  b->alpha.assign(c.algo_alpha, c.algo_alpha + b->S);
  b->beta.assign(b->S, 0);
  if (c.algo_beta) b->beta.assign(c.algo_beta, c.algo_beta + b->S);
  b->synthetic = c.synthetic_exp != 0;
  for (int s = 0; s < b->S; s++) b->any_alpha |= b->alpha[s] != 0;
  /* (the per-flow PF scheduler and the NVS sampler do not use the exponents: dl-pf-packet-scheduler.cpp:128-140, nvs :519-520) */
  for (int s = 0; s < b->S; s++)
    b->gen_exp |= direct && c.sched != RS_SCHED_PF && c.sched != RS_SCHED_NVS_NONGREEDY && ((b->eps[s] | 1) != 1 || (b->psi[s] | 1) != 1);
  b->u2s.assign(c.user_to_slice, c.user_to_slice + b->U);
  b->cfg.cell.slice_weight = nullptr; b->cfg.cell.algo_alpha = nullptr; b->cfg.cell.algo_beta = nullptr;
  b->cfg.cell.algo_epsilon = nullptr; b->cfg.cell.algo_psi = nullptr; b->cfg.cell.user_to_slice = nullptr;
  b->jit_wanted = want_jit && !direct; /* (before the LDS carve: its NVS rule depends on which kernels will run) */
  if (batch_alloc(b)) { rs_batch_destroy(b); return nullptr; }
  if (want_jit && !direct) {
    /* failure is not an error of this call: the built-in kernels stay in use and the batch keeps the reason
     * (rs_batch_jit_status); rs_last_error() is left alone */
    b->jit_wanted = true;
    b->jit = rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, b->sched, 0, slice_window(b), b->jit_msg, sizeof b->jit_msg,
                        b->cfg.cqi_refresh <= 4 ? 2 : 0);
    if (b->jit) snprintf(b->jit_msg, sizeof b->jit_msg, "%s", rs_jit_is_untuned(b->jit) ? "in use, built WITHOUT the -mllvm tuning options (hiprtc refused them): a few per cent slower" : "");
    else if (!b->jit_msg[0]) snprintf(b->jit_msg, sizeof b->jit_msg, "hiprtc build failed");
    if (!b->jit && b->threads > 512) { /* nothing else can launch this workgroup size */
      fail(RS_ERR_INVALID, "threads_per_cell %d needs the shape-specialised kernel, which could not be built: %s", b->threads, b->jit_msg);
      rs_batch_destroy(b);
      return nullptr;
    }
  }
  return b;
}

int check_device_err(rs_batch* b) {
  int e = 0;
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipMemcpy(&e, b->d_err, 4, hipMemcpyDeviceToHost));
  if (e) {
    HIP_TRY(hipMemset(b->d_err, 0, 4));
    return fail(RS_ERR_RANGE, e == RS_CQI_EPOCHS ? "ran past the last CQI epoch" : "trace row outside the uploaded rows");
  }
  return RS_OK;
}

/* Long unlogged runs on epoch grids -- what the measurements and any production batch are -- go to the LEAN build of the
 * shape-specialised kernel: the options such a launch does not use (trace rows, per-PRB twins, the decision log, error-model draws,
 * synthetic-experiment blocks) are compile-time constants there.  Compiled once, at the first launch that qualifies
 * (RS_JIT_LEAN_MIN_TTIS, default 256: short test launches are not worth a hiprtc run; RS_JIT_LEAN=0 switches it off) or by
 * rs_batch_prepare_launch; nullptr = the general kernel (or the built-in ones) serves this launch. */
RsJitKernel* lean_kernel(rs_batch* b, int n_ttis, bool logged) {
  if (!b->jit || b->direct || logged) return nullptr;
  const char* const e_min = getenv("RS_JIT_LEAN_MIN_TTIS");
  const char* const e_on = getenv("RS_JIT_LEAN");
  const int lean_min = e_min ? atoi(e_min) : 256;
  if (e_on && atoi(e_on) == 0) return nullptr;
  if (n_ttis < lean_min || b->cqi_mode != RS_CQI_EPOCHS || b->d_epochs_prb || b->cfg.phy_error_draws || b->synthetic) return nullptr;
  if (!b->jit_lean_tried) {
    b->jit_lean_tried = true;
    char msg[512] = "";
    b->jit_lean = rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, b->sched, b->queues ? b->qmode : 0, slice_window(b), msg,
                             sizeof msg, (b->cfg.cqi_refresh <= 4 ? 2 : 0) | 4);
    /* not an error of the launch -- the general build serves it -- but nothing silent either: rs_batch_jit_status carries the reason */
    if (!b->jit_lean) snprintf(b->jit_msg, sizeof b->jit_msg, "lean build unavailable, the general build serves every launch (%.400s)", msg);
  }
  return b->jit_lean;
}

int launch(rs_batch* b, int n_ttis, int16_t* d_map, int16_t* d_quota, int16_t* d_target, int32_t* d_tbs, int32_t* d_uinfo,
           uint32_t* d_keys = nullptr);

/* Everything a batch carries from one launch to the next, as (device pointer, bytes): PF averages, pending grants, cumulative counters,
 * slice state, clock / rand() ring / CQI-report state, and with the queue model the bearers' queues, averages, counters and per-user
 * words.  One list for the checkpoint (rs_batch_checkpoint_*), the autotune's snapshot and the self-check's. */
struct StatePart { void* p; size_t n; };
std::vector<StatePart> state_parts(rs_batch* b) {
  const size_t cells = b->n_cells, U = b->U, S = b->S, n2 = cells * 2 * U;
  std::vector<StatePart> v = {{b->d_avg, 8 * cells * U}, {b->d_tx, 4 * cells * U}, {b->d_cumb, 8 * cells * U}, {b->d_cumr, 8 * cells * U},
                              {b->d_sstate, 8 * cells * S}, {b->d_scal, sizeof(RsCellScalars) * cells}};
  if (b->queues) {
    v.push_back({b->d_qi, 4 * 7 * n2});
    v.push_back({b->d_bavg, 8 * n2});
    v.push_back({b->d_bcum, 8 * 2 * n2});
    v.push_back({b->d_qflags, cells * U});
    v.push_back({b->d_qhol, 8 * cells * U});
  }
  return v;
}

/* ---- a device-side snapshot of that state: what rs_batch_config.autotune and the self-check put back after their trial launches ---- */
struct StateParts {
  std::vector<StatePart> parts;
  size_t total = 0;
  explicit StateParts(rs_batch* b) : parts(state_parts(b)) {
    for (const StatePart& q : parts) total += (q.n + 255) & ~(size_t)255;
  }
  hipError_t copy(rs_batch* b, void* snapshot, bool save) const {
    size_t off = 0;
    for (const StatePart& q : parts) {
      void* snap = (char*)snapshot + off;
      const hipError_t e = hipMemcpyAsync(save ? snap : q.p, save ? q.p : snap, q.n, hipMemcpyDeviceToDevice, b->stream);
      if (e != hipSuccess) return e;
      off += (q.n + 255) & ~(size_t)255;
    }
    return hipSuccess;
  }
};

/* What a run leaves behind, in the form every kernel of a batch must agree on bit for bit: PF averages, cumulative bytes / RBs, slice
 * state, the bytes of the pending grants (the shape-specialised kernels pack more into that word), clock, rand() ring and CQI-report
 * state.  FNV-1a over the host copy (a few MB, once per trial). */
struct StateDigest {
  unsigned long long part[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; /* PF averages, cumulative bytes, cumulative RBs, slice state, pending grants' bytes, scalars; queue model: the bearers' queues, averages, counters */
  bool operator==(const StateDigest& o) const { return memcmp(part, o.part, sizeof part) == 0; }
  bool operator!=(const StateDigest& o) const { return !(*this == o); }
  std::string diff(const StateDigest& o) const {
    static const char* const names[13] = {"PF averages", "cumulative bytes", "cumulative RBs", "slice state", "pending grants", "simulated time",
                                          "last EWMA update", "TTIs done", "users served in the last TTI", "rand() ring / CQI-report state",
                                          "bearers' queues", "bearers' PF averages", "bearers' cumulative counters"};
    std::string d;
    for (int i = 0; i < 13; i++)
      if (part[i] != o.part[i]) d += std::string(d.empty() ? "" : ", ") + names[i];
    return d;
  }
};
int state_digest(rs_batch* b, StateDigest* out) {
  const size_t cells = b->n_cells, U = b->U, S = b->S;
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::vector<unsigned char> h(8 * cells * U);
  auto fnv = [](const void* p, size_t n, unsigned long long d = 1469598103934665603ull) {
    const unsigned char* c = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { d ^= c[i]; d *= 1099511628211ull; }
    return d;
  };
  int k = 0;
  for (void* dev : {(void*)b->d_avg, (void*)b->d_cumb, (void*)b->d_cumr}) {
    HIP_TRY(hipMemcpy(h.data(), dev, 8 * cells * U, hipMemcpyDeviceToHost));
    out->part[k++] = fnv(h.data(), 8 * cells * U);
  }
  HIP_TRY(hipMemcpy(h.data(), b->d_sstate, 8 * cells * S, hipMemcpyDeviceToHost));
  out->part[3] = fnv(h.data(), 8 * cells * S);
  std::vector<int32_t> tx(cells * U);
  HIP_TRY(hipMemcpy(tx.data(), b->d_tx, 4 * cells * U, hipMemcpyDeviceToHost));
  for (int32_t& v : tx) v &= RS_TX_BYTES_MASK;
  out->part[4] = fnv(tx.data(), 4 * cells * U);
  std::vector<RsCellScalars> sc(cells);
  HIP_TRY(hipMemcpy(sc.data(), b->d_scal, sizeof(RsCellScalars) * cells, hipMemcpyDeviceToHost));
  for (int i = 5; i < 10; i++) out->part[i] = 1469598103934665603ull;
  for (const RsCellScalars& c : sc) {
    out->part[5] = fnv(&c.t, 8, out->part[5]);
    out->part[6] = fnv(&c.last_update, 8, out->part[6]);
    out->part[7] = fnv(&c.n_done, 8, out->part[7]);
    /* (read by the error model's draws only: a lean build, which serves batches without them, does not keep it) */
    if (b->cfg.phy_error_draws) out->part[8] = fnv(&c.served_prev, 4, out->part[8]);
    unsigned long long d = out->part[9];
    if (b->cqi_mode == RS_CQI_TRACE) { d = fnv(&c.last_sent, 8, d); d = fnv(&c.reported, 4, d); d = fnv(&c.cqi_row, 4, d); }
    /* the ring in age order: the kernels may leave it rotated differently (f, b) with the same future */
    for (int i = 0; i < 31; i++) { const uint32_t w = c.rng_r[(c.rng_f + i) % 31]; d = fnv(&w, 4, d); }
    out->part[9] = d;
  }
  if (b->queues) {
    /* the queue model: head, tail, pk, frag, bytes, pkts, tx of every bearer; its PF average; its cumulative bytes / RBs.  (The per-user
     * flag and HoL words are rebuilt by every TTI's first phase before anything reads them: scratch, not compared.) */
    const size_t n2 = cells * 2 * U;
    const StatePart q[3] = {{b->d_qi, 4 * 7 * n2}, {b->d_bavg, 8 * n2}, {b->d_bcum, 8 * 2 * n2}};
    for (int i = 0; i < 3; i++) {
      std::vector<unsigned char> hq(q[i].n);
      HIP_TRY(hipMemcpy(hq.data(), q[i].p, q[i].n, hipMemcpyDeviceToHost));
      out->part[10 + i] = fnv(hq.data(), q[i].n);
    }
  }
  return RS_OK;
}

/* rs_batch_config.autotune: time the lean kernel in the variants the rule table of rs_jit.cpp chooses between, on this batch's own
 * next TTIs, and keep the fastest.  Every trial starts from the same snapshot and the snapshot is put back at the end, so the tuning
 * leaves no trace in the run.  All variants compute the same results (the parity tests run them against the oracle) -- and every
 * trial's final state is compared with the rule table's: a variant that disagrees is never kept (and says so in the report). */
int autotune(rs_batch* b, int n_ttis) {
  if (b->autotuned || !b->cfg.autotune) return RS_OK;
  if (!lean_kernel(b, n_ttis, false)) return RS_OK; /* this launch does not qualify for the lean build: try again at a later one */
  b->autotuned = true;
  const int sched = b->sched;
  if (b->queues || b->direct || !(sched == 8 || sched == 9 || sched == 101 || sched == 103)) return RS_OK;
  const int win = slice_window(b);
  std::vector<std::string> cand = {""};
  const bool ilp_by_rule = sched == 9 || (sched == 8 && (b->R > 32 || win > 0));
  cand.push_back(ilp_by_rule ? "ss=default" : "ss=iterative-ilp");
  cand.push_back(b->R > 32 ? "-DRS_NO_SPEC" : "-DRS_NO_HOLD");
  if (sched == 9) cand.push_back("-DRS_P3_BLOCK=8");
  const StateParts sp(b);
  /* the snapshot and the two events live exactly as long as this call, whichever way it returns (an early HIP_TRY return used
   * to leave them -- and the candidate kernel -- behind) */
  ScratchBuffers scratch;
  ScratchEvents events;
  void* snapshot = nullptr;
  RsJitKernel* const k_default = b->jit_lean;
  const int64_t done0 = b->ttis_done;
  struct Restore { /* whatever happens: the rule table's build unless a better one was chosen, the TTI count as it was */
    rs_batch* b; RsJitKernel* k; int64_t done;
    ~Restore() { b->jit_lean = k; b->ttis_done = done; }
  } restore{b, k_default, done0};
  HIP_TRY(scratch.alloc(&snapshot, sp.total));
  HIP_TRY(events.create(2));
  const hipEvent_t e0 = events.ev[0], e1 = events.ev[1];
  const int trial = n_ttis < 512 ? n_ttis : 512;
  RsJitKernel* best = k_default;
  float best_ms = 0, default_ms = 0;
  StateDigest default_digest;
  int rc = RS_OK;
  HIP_TRY(sp.copy(b, snapshot, true));
  std::string report;
  for (size_t i = 0; i < cand.size() && rc == RS_OK; i++) {
    char msg[512] = "";
    RsJitKernel* k = i == 0 ? k_default
                            : rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, sched, 0, win, msg, sizeof msg,
                                         (b->cfg.cqi_refresh <= 4 ? 2 : 0) | 4, cand[i].c_str());
    if (!k) { report += cand[i] + ": not built; "; continue; }
    b->jit_lean = k;
    float ms = 0;
    for (int rep = 0; rep < 3 && rc == RS_OK; rep++) { /* one warm launch, then the faster of two */
      if (sp.copy(b, snapshot, false) != hipSuccess || hipEventRecord(e0, b->stream) != hipSuccess) { rc = fail(RS_ERR_HIP, "autotune: state restore failed"); break; }
      rc = launch(b, trial, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      if (rc) break;
      float t = 0;
      if (hipEventRecord(e1, b->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&t, e0, e1) != hipSuccess) {
        rc = fail(RS_ERR_HIP, "autotune: timing failed");
        break;
      }
      if (rep == 1 || (rep == 2 && t < ms)) ms = t;
    }
    if (rc) break;
    StateDigest digest;
    rc = state_digest(b, &digest);
    if (rc) break;
    if (i == 0) default_digest = digest;
    char line[320];
    if (digest != default_digest) {
      /* never seen; if it ever is, the variant is a miscompiled kernel (profiles/r05_onelane.md) and must not serve the batch */
      snprintf(line, sizeof line, "%s: %.3f ms, REJECTED: its state after the trial differs from the rule table's build (%s); ", cand[i].c_str(), ms,
               digest.diff(default_digest).c_str());
      report += line;
      rs_jit_reject(k);
      continue;
    }
    /* a variant that agrees with a build that carries the self-check mark has passed the same check by transitivity */
    if (i > 0 && rs_jit_is_verified(k_default)) rs_jit_mark_verified(k);
    snprintf(line, sizeof line, "%s: %.3f ms; ", i == 0 ? "rule table" : cand[i].c_str(), ms);
    report += line;
    b->autotune_n++;
    /* (another build has to win by more than 1 %: below that the launches' own jitter decides) */
    if (i == 0) { best = k; best_ms = default_ms = ms; }
    else if (ms < best_ms && ms < 0.99f * default_ms) { best = k; best_ms = ms; }
  }
  /* the run continues from where it was, whatever happened above */
  const hipError_t back = sp.copy(b, snapshot, false);
  const hipError_t sync = hipStreamSynchronize(b->stream);
  if (back != hipSuccess || sync != hipSuccess) return fail(RS_ERR_HIP, "autotune: the cell state could not be put back");
  if (rc) return rc;
  rc = check_device_err(b);
  if (rc) return rc;
  restore.k = best;
  snprintf(b->autotune_msg, sizeof b->autotune_msg, "autotune over %d TTIs: %skept %s", trial, report.c_str(),
           best == k_default ? "the rule table's build" : "the fastest");
  return RS_OK;
}

/* Which run-time builds a batch checks against the built-in kernels before it trusts them (round 6: on by default).
 *   rs_batch_config.selfcheck  1: every build, marked or not;  -1: none;  0 (default): every build WITHOUT the self-check mark -- one
 *   that this process compiled, or that came from the disk cache of a process that never checked it.  A build that passed leaves the mark
 *   in its cache file (rs_jit_mark_verified), so the second process of a campaign skips the check.
 *   RS_JIT_SELFCHECK=0 switches the default off (cfg.selfcheck = 1 still checks), RS_JIT_SELFCHECK=2 makes it "every build". */
enum { kCheckNever = 0, kCheckUnmarked = 1, kCheckAll = 2 };
int selfcheck_policy(const rs_batch* b) {
  if (b->cfg.selfcheck > 0) return kCheckAll;
  if (b->cfg.selfcheck < 0) return kCheckNever;
  const char* e = getenv("RS_JIT_SELFCHECK");
  if (e && e[0] == '0') return kCheckNever;
  if (e && e[0] == '2') return kCheckAll;
  return kCheckUnmarked;
}

/* The self-check: the batch's next min(n, 256) TTIs on the kernels built into the library and on its run-time compiled ones (general
 * build, lean build), each from the same snapshot; the final states must agree bit for bit (state_digest).  The built-in kernels are
 * ONE binary, the one the GPU parity suite runs against the oracle; a run-time build is a fresh compilation for this shape, and round 4
 * met one that the compiler got wrong (profiles/r05_onelane.md).  A general build that disagrees is dropped: the batch runs on the
 * built-in kernels and rs_batch_jit_status returns -2 with the reason; a lean build that disagrees while the general one agreed is
 * dropped alone (status stays 1, the message says so).  The snapshot is put back: no trace.  With the queue model the snapshot and the
 * comparison include the bearers' queues, averages and counters (they were left out and the check refused nothing but
 * silently advanced them). */
int selfcheck(rs_batch* b, int n_ttis, bool logged) {
  const int policy = selfcheck_policy(b);
  if (policy == kCheckNever || b->selfchecked || b->in_selfcheck) return RS_OK;
  if (!b->jit || b->direct) { b->selfchecked = true; return RS_OK; } /* nothing run-time compiled */
  if (b->threads > 512) { /* no built-in kernel of this workgroup size to compare with */
    b->selfchecked = true;
    snprintf(b->selfcheck_msg, sizeof b->selfcheck_msg, "not self-checked: the kernels built into the library stop at 512 threads per cell");
    return RS_OK;
  }
  /* the general build at the first launch; the lean build at the first launch that qualifies for it (RS_JIT_LEAN_MIN_TTIS, unlogged) */
  RsJitKernel* const k_gen = b->jit;
  RsJitKernel* const k_lean = lean_kernel(b, n_ttis, logged);
  const bool need_gen = !b->selfchecked_general && (policy == kCheckAll || !rs_jit_is_verified(k_gen));
  const bool need_lean = k_lean && !b->selfchecked_lean && (policy == kCheckAll || !rs_jit_is_verified(k_lean));
  const char* const e_on = getenv("RS_JIT_LEAN");
  const bool lean_off = e_on && atoi(e_on) == 0;
  if (!need_gen && !need_lean) {
    /* builds that came with the mark of an earlier process's check */
    if (!b->selfchecked_general) {
      b->selfchecked_general = true;
      snprintf(b->selfcheck_msg, sizeof b->selfcheck_msg, "the run-time build carries the self-check mark of the process that compiled it (cache file)");
    }
    if (k_lean) b->selfchecked_lean = true;
    b->selfchecked = b->selfchecked_lean || lean_off;
    return RS_OK;
  }
  b->in_selfcheck = true;
  struct Leave { rs_batch* b; ~Leave() { b->in_selfcheck = false; } } leave{b};
  const StateParts sp(b);
  ScratchBuffers scratch;
  void* snap = nullptr;
  HIP_TRY(scratch.alloc(&snap, sp.total));
  const int64_t done0 = b->ttis_done;
  const int trial = n_ttis < 256 ? n_ttis : 256;
  int rc = RS_OK;
  StateDigest want, got;
  const char* bad = nullptr;
  /* the pending-grant words of a shape-specialised kernel are bytes | PRBs << 20 | "counted" << 30 (RS_TX_*: the grant is in the
   * cumulative totals already); the built-in kernels keep plain bytes there (and count at grant time).  The queue model's kernels, built-in
   * and run-time, share one format (and consume every grant before the launch ends): nothing to convert there. */
  auto plain_tx = [&]() -> int {
    if (b->queues) return RS_OK;
    std::vector<int32_t> tx((size_t)b->n_cells * b->U);
    if (hipStreamSynchronize(b->stream) != hipSuccess || hipMemcpy(tx.data(), b->d_tx, 4 * tx.size(), hipMemcpyDeviceToHost) != hipSuccess) return fail(RS_ERR_HIP, "selfcheck: copy failed");
    for (int32_t& w : tx) w &= RS_TX_BYTES_MASK;
    if (hipMemcpy(b->d_tx, tx.data(), 4 * tx.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(RS_ERR_HIP, "selfcheck: copy failed");
    return RS_OK;
  };
  if (sp.copy(b, snap, true) != hipSuccess) return fail(RS_ERR_HIP, "selfcheck: snapshot failed");
  for (int v = 0; v < 3 && rc == RS_OK && !bad; v++) {
    /* v = 0: built-in (no run-time kernel visible to launch()), 1: the general build alone, 2: the lean build */
    if ((v == 1 && !need_gen) || (v == 2 && !need_lean)) continue;
    b->jit = v == 0 ? nullptr : k_gen;
    RsJitKernel* const keep_lean = b->jit_lean;
    const bool keep_tried = b->jit_lean_tried;
    if (v < 2) { b->jit_lean = nullptr; b->jit_lean_tried = true; }
    if (sp.copy(b, snap, false) != hipSuccess) rc = fail(RS_ERR_HIP, "selfcheck: state restore failed");
    if (!rc && v == 0) rc = plain_tx();
    if (!rc) rc = launch(b, trial, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    b->jit_lean = keep_lean;
    b->jit_lean_tried = keep_tried;
    if (!rc) rc = state_digest(b, v == 0 ? &want : &got);
    if (!rc && v > 0 && got != want) bad = v == 1 ? "general" : "lean";
  }
  b->jit = k_gen;
  const hipError_t back = sp.copy(b, snap, false);
  const hipError_t sync = hipStreamSynchronize(b->stream);
  b->ttis_done = done0;
  if (back != hipSuccess || sync != hipSuccess) return fail(RS_ERR_HIP, "selfcheck: the cell state could not be put back");
  if (rc) return rc;
  rc = check_device_err(b);
  if (rc) return rc;
  if (bad && bad[0] == 'l') {
    /* the lean build alone is wrong (the general one agreed just now or earlier, or carries the mark): the general build keeps serving every launch */
    rs_jit_reject(k_lean);
    if (need_gen) { b->selfchecked_general = true; rs_jit_mark_verified(k_gen); }
    b->jit_lean = nullptr;
    b->jit_lean_tried = true;
    b->selfchecked_lean = true;
    b->selfchecked = true;
    snprintf(b->jit_msg, sizeof b->jit_msg, "selfcheck: after %d TTIs the lean build of the shape-specialised kernel left a state that differs from the "
             "built-in kernels' in: %s; it is dropped and the general build, which agreed, serves every launch", trial, got.diff(want).c_str());
    return RS_OK;
  }
  if (bad) {
    rs_jit_reject(k_gen);
    b->jit = nullptr;
    b->jit_lean = nullptr;
    b->jit_lean_tried = true;
    b->jit_rejected = true;
    b->selfchecked = true;
    /* the state just put back may have been left by the shape-specialised kernel: the built-in kernels that take over read plain bytes
     * (the first EWMA after a rejection used the packed word as a byte count) */
    rc = plain_tx();
    if (rc) return rc;
    snprintf(b->jit_msg, sizeof b->jit_msg, "selfcheck: after %d TTIs the %s build of the shape-specialised kernel left a state that differs from the "
             "built-in kernels' in: %s; the batch runs on the built-in kernels (lint the code object: tools/lint_exec_restore.py)", trial, bad,
             got.diff(want).c_str());
    return RS_OK;
  }
  if (need_gen) { b->selfchecked_general = true; rs_jit_mark_verified(k_gen); }
  if (need_lean) { b->selfchecked_lean = true; rs_jit_mark_verified(k_lean); }
  if (!need_lean && k_lean) b->selfchecked_lean = true; /* (came with the mark) */
  snprintf(b->selfcheck_msg, sizeof b->selfcheck_msg, "selfcheck over %d TTIs: the built-in kernels and the %s build%s agree", trial,
           b->selfchecked_lean ? "general and lean" : "general", b->selfchecked_lean ? "s" : "");
  b->selfchecked = b->selfchecked_lean || lean_off; /* the lean build is still to come: look again at the next launch */
  return RS_OK;
}

int launch(rs_batch* b, int n_ttis, int16_t* d_map, int16_t* d_quota, int16_t* d_target, int32_t* d_tbs, int32_t* d_uinfo,
           uint32_t* d_keys) {
  if (n_ttis < 1) return fail(RS_ERR_INVALID, "n_ttis %d < 1", n_ttis);
  if (b->cqi_mode == RS_CQI_NONE) return fail(RS_ERR_STATE, "no CQI source set");
  if (b->any_alpha && !b->direct && !b->queues)
    return fail(RS_ERR_STATE, "customised slices (algo_alpha = 1) read queue state: call rs_batch_set_bearers / rs_batch_set_arrivals first");
  if (b->queues && !b->d_arr_off) return fail(RS_ERR_STATE, "rs_batch_set_arrivals has not been called");
  if (n_ttis > RS_MAX_TTIS_PER_LAUNCH) {
    /* longer runs go out as several launches (state carries over; the per-launch counters of the kernel are 32-bit) */
    if (d_map || d_tbs || d_uinfo || d_keys) return fail(RS_ERR_INVALID, "logged runs are limited to %d TTIs per call", RS_MAX_TTIS_PER_LAUNCH);
    for (int done = 0; done < n_ttis;) {
      const int chunk = n_ttis - done < RS_MAX_TTIS_PER_LAUNCH ? n_ttis - done : RS_MAX_TTIS_PER_LAUNCH;
      const int rc = launch(b, chunk, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
      if (rc) return rc;
      done += chunk;
    }
    return RS_OK;
  }
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  RsLaunch L = b->base;
  L.n_ttis = n_ttis;
  L.cqi_mode = b->cqi_mode;
  L.epochs = b->d_epochs; L.grid_stride = b->grid_stride; L.n_epochs = b->n_epochs;
  L.trace = b->d_trace; L.n_traces = b->n_traces; L.n_rows = b->n_rows; L.row_mod = b->row_mod;
  L.user_trace = b->d_user_trace;
  L.epochs_prb = b->cqi_mode == RS_CQI_EPOCHS ? b->d_epochs_prb : nullptr; L.grid_stride_prb = b->grid_stride_prb;
  L.trace_prb = b->cqi_mode == RS_CQI_TRACE ? b->d_trace_prb : nullptr;
  L.log_map = d_map; L.log_quota = d_quota; L.log_target = d_target; L.log_tbs = d_tbs; L.log_uinfo = d_uinfo;
  L.log_keys = d_keys;
  if (b->queues) {
    const size_t n = (size_t)b->n_cells * 2 * b->U;
    L.bearer_kind = b->d_bearer_kind; L.arr_off = b->d_arr_off; L.arr_time = b->d_arr_time;
    L.arr_nfull = b->d_arr_nfull; L.arr_last = b->d_arr_last;
    L.q_head = b->d_qi; L.q_tail = b->d_qi + n; L.q_pk = b->d_qi + 2 * n; L.q_frag = b->d_qi + 3 * n;
    L.q_bytes = b->d_qi + 4 * n; L.q_pkts = b->d_qi + 5 * n; L.b_tx = b->d_qi + 6 * n;
    L.b_avg = b->d_bavg; L.b_cumb = b->d_bcum; L.b_cumr = b->d_bcum + n;
    L.q_flags = b->d_qflags; L.q_hol = b->d_qhol;
  }
  const bool logged = d_map || d_quota || d_target || d_tbs || d_uinfo || d_keys;
  if (!b->selfchecked && !b->in_selfcheck && b->jit && !b->direct) {
    const int rc = selfcheck(b, n_ttis, logged); /* (its trials come back through here with `in_selfcheck` set) */
    if (rc) return rc;
  }
  if (b->cfg.autotune && !b->autotuned && !logged && !b->in_selfcheck) { /* (never inside a self-check trial: it checks the rule table's build) */
    const int rc = autotune(b, n_ttis);
    if (rc) return rc;
  }
  if (L.prio_sum) HIP_TRY(hipMemsetAsync(L.prio_sum, 0, 8, b->stream)); /* (every launch counts from zero) */
  RsJitKernel* k = lean_kernel(b, n_ttis, logged);
  if (!k) k = b->jit;
  if (k) HIP_TRY(rs_jit_launch(k, &L, b->stream));
  else HIP_TRY(rs_launch_cells(&L, b->threads, b->stream));
  b->ttis_done += n_ttis;
  return RS_OK;
}

}  // namespace

extern "C" {

rs_batch* rs_batch_create(const rs_batch_config* cfg) { return batch_new(cfg, false); }

/* the same behind a deterministic layout check: the caller passes the ABI version and the struct size IT was compiled with
 * (RS_BATCH_CREATE in the header does); a caller built against another layout is refused instead of being read field-shifted */
rs_batch* rs_batch_create_checked(const rs_batch_config* cfg, int abi_version, size_t cfg_size) {
  if (abi_version != RS_ABI_VERSION || cfg_size != sizeof(rs_batch_config)) {
    fail(RS_ERR_INVALID, "ABI mismatch: caller was built against ABI %d with an rs_batch_config of %zu bytes, this library is ABI %d with %zu bytes",
         abi_version, cfg_size, RS_ABI_VERSION, sizeof(rs_batch_config));
    return nullptr;
  }
  return rs_batch_create(cfg);
}

void rs_batch_destroy(rs_batch* b) {
  if (!b) return;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  void* ptrs[] = {b->d_tab, b->d_weight, b->d_eps, b->d_psi, b->d_alpha, b->d_beta, b->d_user_slice, b->d_tbs_eff, b->d_avg, b->d_tx, b->d_cumb, b->d_cumr,
                  b->d_sstate, b->d_scal, b->d_epochs, b->d_trace, b->d_user_trace, b->d_err, b->d_slice_bytes, b->d_stamps,
                  b->d_bearer_kind, b->d_arr_off, b->d_arr_time, b->d_arr_nfull, b->d_arr_last, b->d_qi, b->d_bavg, b->d_bcum,
                  b->d_qflags, b->d_qhol, b->d_epochs_prb, b->d_trace_prb, b->d_gen_num, b->d_prio_sum};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

int rs_batch_seed(rs_batch* b, const uint32_t* seed, const int64_t* rand_skip) {
  if (!b || !seed) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  if (b->ttis_done) return fail(RS_ERR_STATE, "rs_batch_seed after TTIs were run");
  return init_scalars(b, seed, rand_skip);
}

int rs_batch_upload_cqi_epochs(rs_batch* b, const uint8_t* h_cqi, int32_t n_epochs) {
  if (!b || !h_cqi || n_epochs < 1) return fail(RS_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  const size_t grid = (size_t)b->U * b->R;
  /* device-resident form = the cell kernel's LDS image: RBG-major [R][Upad] with zero padding (a refresh is a straight copy) */
  const size_t Upad = (size_t)rs_upad_of(b->U);
  const size_t stride = round_up((int)(Upad * b->R), 16);
  const size_t total = (size_t)b->n_cells * n_epochs;
  for (size_t i = 0; i < total * grid; i++)
    if (h_cqi[i] < 1 || h_cqi[i] > 15) return fail(RS_ERR_INVALID, "CQI %d at byte %zu outside 1..15", h_cqi[i], i);
  std::vector<uint8_t> packed(total * stride, 0);
  for (size_t g = 0; g < total; g++)
    for (size_t u = 0; u < (size_t)b->U; u++)
      for (size_t r = 0; r < (size_t)b->R; r++) packed[g * stride + r * Upad + u] = h_cqi[g * grid + u * b->R + r];
  if (b->d_epochs) { HIP_TRY(hipFree(b->d_epochs)); b->d_epochs = nullptr; }
  HIP_TRY(hipMalloc(&b->d_epochs, packed.size()));
  HIP_TRY(hipMemcpy(b->d_epochs, packed.data(), packed.size(), hipMemcpyHostToDevice));
  b->grid_stride = (int64_t)stride;
  b->n_epochs = n_epochs;
  b->cqi_mode = RS_CQI_EPOCHS;
  if (b->d_epochs_prb) { HIP_TRY(hipFree(b->d_epochs_prb)); b->d_epochs_prb = nullptr; }
  return RS_OK;
}

int rs_batch_upload_cqi_epochs_prb(rs_batch* b, const uint8_t* h_cqi_prb, int32_t n_epochs) {
  if (!b || !h_cqi_prb || n_epochs < 1) return fail(RS_ERR_INVALID, "bad argument");
  const size_t per_user = (size_t)b->R * b->G, grid = (size_t)b->U * per_user, total = (size_t)b->n_cells * n_epochs;
  /* the metric reads the first PRB of every RBG (ref: downlink-transport-scheduler.cpp:536): the per-RBG grids the kernels keep in LDS */
  std::vector<uint8_t> rbg(total * b->U * b->R);
  for (size_t g = 0; g < total; g++)
    for (size_t u = 0; u < (size_t)b->U; u++)
      for (size_t r = 0; r < (size_t)b->R; r++) rbg[(g * b->U + u) * b->R + r] = h_cqi_prb[g * grid + u * per_user + r * b->G];
  for (size_t i = 0; i < total * grid; i++)
    if (h_cqi_prb[i] < 1 || h_cqi_prb[i] > 15) return fail(RS_ERR_INVALID, "CQI %d at byte %zu outside 1..15", h_cqi_prb[i], i);
  int rc = rs_batch_upload_cqi_epochs(b, rbg.data(), n_epochs);
  if (rc) return rc;
  const size_t stride = round_up((int)grid, 16);
  std::vector<uint8_t> packed(total * stride, 0);
  for (size_t g = 0; g < total; g++) memcpy(&packed[g * stride], h_cqi_prb + g * grid, grid);
  HIP_TRY(hipMalloc(&b->d_epochs_prb, packed.size()));
  HIP_TRY(hipMemcpy(b->d_epochs_prb, packed.data(), packed.size(), hipMemcpyHostToDevice));
  b->grid_stride_prb = (int64_t)stride;
  return RS_OK;
}

int rs_batch_synthesize_cqi(rs_batch* b, uint64_t seed, const double* w, int32_t n_epochs) {
  return rs_batch_synthesize_cqi_at(b, seed, w, n_epochs, 0);
}

int rs_batch_synthesize_cqi_at(rs_batch* b, uint64_t seed, const double* w, int32_t n_epochs, int64_t first_cell) {
  if (!b || !w || n_epochs < 1 || first_cell < 0) return fail(RS_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  double tot = 0;
  for (int i = 0; i < 15; i++) {
    if (!(w[i] >= 0)) return fail(RS_ERR_INVALID, "negative CQI weight");
    tot += w[i];
  }
  if (!(tot > 0)) return fail(RS_ERR_INVALID, "all CQI weights are zero");
  uint32_t cdf[16];
  double acc = 0;
  for (int i = 0; i < 15; i++) {
    acc += w[i];
    double v = floor(acc / tot * 4294967296.0);
    cdf[i] = v >= 4294967295.0 ? 4294967295u : (uint32_t)v;
  }
  cdf[14] = 4294967295u;
  cdf[15] = 4294967295u;
  const int Upad = rs_upad_of(b->U);
  const size_t stride = round_up(Upad * b->R, 16); /* RBG-major [R][Upad]: the cell kernel's LDS image */
  const size_t bytes = (size_t)b->n_cells * n_epochs * stride;
  if (b->d_epochs) { HIP_TRY(hipFree(b->d_epochs)); b->d_epochs = nullptr; }
  HIP_TRY(hipMalloc(&b->d_epochs, bytes));
  HIP_TRY(rs_launch_synth(b->d_epochs, (int64_t)stride, b->n_cells, n_epochs, b->U, b->R, Upad, seed, first_cell, cdf, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->grid_stride = (int64_t)stride;
  b->n_epochs = n_epochs;
  b->cqi_mode = RS_CQI_EPOCHS;
  /* the per-PRB twin of an earlier upload belongs to the grids just replaced (other epoch count, other values) */
  if (b->d_epochs_prb) { HIP_TRY(hipFree(b->d_epochs_prb)); b->d_epochs_prb = nullptr; }
  b->grid_stride_prb = 0;
  return RS_OK;
}

int rs_batch_download_cqi_epochs(rs_batch* b, int32_t cell, uint8_t* h_cqi) {
  if (!b || !h_cqi || cell < 0 || cell >= b->n_cells) return fail(RS_ERR_INVALID, "bad argument");
  if (b->cqi_mode != RS_CQI_EPOCHS) return fail(RS_ERR_STATE, "no epoch grids on the device");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const size_t grid = (size_t)b->U * b->R, stride = (size_t)b->grid_stride;
  std::vector<uint8_t> tmp((size_t)b->n_epochs * stride);
  HIP_TRY(hipMemcpy(tmp.data(), b->d_epochs + (size_t)cell * b->n_epochs * stride, tmp.size(), hipMemcpyDeviceToHost));
  const size_t Upad = (size_t)rs_upad_of(b->U);
  for (int e = 0; e < b->n_epochs; e++) /* back to the caller's [U][R] order */
    for (size_t u = 0; u < (size_t)b->U; u++)
      for (size_t r = 0; r < (size_t)b->R; r++) h_cqi[e * grid + u * b->R + r] = tmp[e * stride + r * Upad + u];
  return RS_OK;
}

int rs_batch_set_trace(rs_batch* b, const uint8_t* h_trace, int32_t n_traces, int32_t n_rows, int32_t row_modulus,
                       const int32_t* h_user_trace) {
  if (!b || !h_trace || !h_user_trace || n_traces < 1 || n_rows < 1 || row_modulus < 1)
    return fail(RS_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  const size_t tb = (size_t)n_traces * n_rows * b->R;
  for (size_t i = 0; i < tb; i++)
    if (h_trace[i] < 1 || h_trace[i] > 15) return fail(RS_ERR_INVALID, "trace CQI %d outside 1..15", h_trace[i]);
  const size_t nu = (size_t)b->n_cells * b->U;
  for (size_t i = 0; i < nu; i++)
    if (h_user_trace[i] < 0 || h_user_trace[i] >= n_traces) return fail(RS_ERR_INVALID, "trace id %d out of range", h_user_trace[i]);
  if (b->d_trace) { HIP_TRY(hipFree(b->d_trace)); b->d_trace = nullptr; }
  if (b->d_user_trace) { HIP_TRY(hipFree(b->d_user_trace)); b->d_user_trace = nullptr; }
  HIP_TRY(hipMalloc(&b->d_trace, tb));
  HIP_TRY(hipMemcpy(b->d_trace, h_trace, tb, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&b->d_user_trace, 4 * nu));
  HIP_TRY(hipMemcpy(b->d_user_trace, h_user_trace, 4 * nu, hipMemcpyHostToDevice));
  b->n_traces = n_traces; b->n_rows = n_rows; b->row_mod = row_modulus;
  b->cqi_mode = RS_CQI_TRACE;
  if (b->d_trace_prb) { HIP_TRY(hipFree(b->d_trace_prb)); b->d_trace_prb = nullptr; }
  return RS_OK;
}

int rs_batch_set_trace_prb(rs_batch* b, const uint8_t* h_trace_prb, int32_t n_traces, int32_t n_rows, int32_t row_modulus,
                           const int32_t* h_user_trace) {
  if (!b || !h_trace_prb || n_traces < 1 || n_rows < 1) return fail(RS_ERR_INVALID, "bad argument");
  const size_t per_row = (size_t)b->R * b->G, rows = (size_t)n_traces * n_rows;
  for (size_t i = 0; i < rows * per_row; i++)
    if (h_trace_prb[i] < 1 || h_trace_prb[i] > 15) return fail(RS_ERR_INVALID, "trace CQI %d outside 1..15", h_trace_prb[i]);
  std::vector<uint8_t> rbg(rows * b->R);
  for (size_t q = 0; q < rows; q++)
    for (size_t r = 0; r < (size_t)b->R; r++) rbg[q * b->R + r] = h_trace_prb[q * per_row + r * b->G];
  int rc = rs_batch_set_trace(b, rbg.data(), n_traces, n_rows, row_modulus, h_user_trace);
  if (rc) return rc;
  HIP_TRY(hipMalloc(&b->d_trace_prb, rows * per_row));
  HIP_TRY(hipMemcpy(b->d_trace_prb, h_trace_prb, rows * per_row, hipMemcpyHostToDevice));
  return RS_OK;
}

int rs_batch_run_async(rs_batch* b, int32_t n_ttis) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  return launch(b, n_ttis, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int rs_batch_sync(rs_batch* b) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return check_device_err(b);
}

int rs_batch_run(rs_batch* b, int32_t n_ttis) {
  int rc = rs_batch_run_async(b, n_ttis);
  if (rc) return rc;
  return rs_batch_sync(b);
}

int rs_batch_run_logged(rs_batch* b, int32_t n_ttis, int16_t* h_map, int32_t* h_tbs, int16_t* h_quota,
                        int16_t* h_target, int32_t* h_uinfo) {
  rs_batch_log lg;
  memset(&lg, 0, sizeof lg);
  lg.rbg_to_user = h_map; lg.tbs_bits = h_tbs; lg.quota = h_quota; lg.target = h_target; lg.uinfo = h_uinfo;
  return rs_batch_run_logged_ex(b, n_ttis, &lg);
}

int rs_batch_run_logged_ex(rs_batch* b, int32_t n_ttis, const rs_batch_log* lg) {
  if (!b || !lg) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  if (n_ttis < 1) return fail(RS_ERR_INVALID, "n_ttis %d < 1", n_ttis);
  const bool transport = b->sched == RS_SCHED_SEQUENTIAL || b->sched == RS_SCHED_MAXCELL || b->sched == RS_SCHED_UPPERBOUND ||
                         b->sched == RS_SCHED_SUBOPT || b->sched == RS_SCHED_VOGEL;
  if (lg->slice_keys && !transport) return fail(RS_ERR_INVALID, "slice_keys: only the DownlinkTransportScheduler policies have them");
  const size_t rows = (size_t)b->n_cells * n_ttis;
  int16_t *d_map = nullptr, *d_quota = nullptr, *d_target = nullptr;
  int32_t *d_tbs = nullptr, *d_uinfo = nullptr;
  uint32_t* d_keys = nullptr;
  ScratchBuffers scratch;
  HIP_TRY(scratch.alloc(&d_map, 2 * rows * b->R));
  HIP_TRY(scratch.alloc(&d_quota, 2 * rows * b->S));
  HIP_TRY(scratch.alloc(&d_target, 2 * rows * b->S));
  HIP_TRY(scratch.alloc(&d_tbs, 4 * rows * b->U));
  HIP_TRY(scratch.alloc(&d_uinfo, 4 * rows * b->U));
  if (lg->slice_keys) HIP_TRY(scratch.alloc(&d_keys, 4 * rows * b->R * b->S));
  HIP_TRY(hipMemsetAsync(d_tbs, 0, 4 * rows * b->U, b->stream));
  HIP_TRY(hipMemsetAsync(d_uinfo, 0, 4 * rows * b->U, b->stream));
  int rc = launch(b, n_ttis, d_map, d_quota, d_target, d_tbs, d_uinfo, d_keys);
  if (!rc) rc = rs_batch_sync(b);
  if (!rc) {
    if (lg->rbg_to_user) HIP_TRY(hipMemcpy(lg->rbg_to_user, d_map, 2 * rows * b->R, hipMemcpyDeviceToHost));
    if (lg->quota) HIP_TRY(hipMemcpy(lg->quota, d_quota, 2 * rows * b->S, hipMemcpyDeviceToHost));
    if (lg->target) HIP_TRY(hipMemcpy(lg->target, d_target, 2 * rows * b->S, hipMemcpyDeviceToHost));
    if (lg->tbs_bits) HIP_TRY(hipMemcpy(lg->tbs_bits, d_tbs, 4 * rows * b->U, hipMemcpyDeviceToHost));
    if (lg->uinfo) HIP_TRY(hipMemcpy(lg->uinfo, d_uinfo, 4 * rows * b->U, hipMemcpyDeviceToHost));
    if (lg->slice_keys) HIP_TRY(hipMemcpy(lg->slice_keys, d_keys, 4 * rows * b->R * b->S, hipMemcpyDeviceToHost));
  }
  return rc;
}

int rs_batch_run_timed(rs_batch* b, int32_t n_ttis, int32_t launches, float* ms) {
  if (!b || !ms || launches < 1) return fail(RS_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  ScratchEvents events;
  HIP_TRY(events.create(launches + 1));
  std::vector<hipEvent_t>& ev = events.ev;
  int rc = RS_OK;
  HIP_TRY(hipEventRecord(ev[0], b->stream));
  for (int i = 0; i < launches && !rc; i++) {
    rc = launch(b, n_ttis, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (!rc) HIP_TRY(hipEventRecord(ev[i + 1], b->stream));
  }
  if (!rc) rc = rs_batch_sync(b);
  if (!rc)
    for (int i = 0; i < launches; i++) HIP_TRY(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
  return rc;
}

int rs_batch_read_state(rs_batch* b, double* avg, int64_t* cum_bytes, int64_t* cum_rbs, double* slice_state) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const size_t n = (size_t)b->n_cells * b->U;
  if (avg) HIP_TRY(hipMemcpy(avg, b->d_avg, 8 * n, hipMemcpyDeviceToHost));
  if (cum_bytes) HIP_TRY(hipMemcpy(cum_bytes, b->d_cumb, 8 * n, hipMemcpyDeviceToHost));
  if (cum_rbs) HIP_TRY(hipMemcpy(cum_rbs, b->d_cumr, 8 * n, hipMemcpyDeviceToHost));
  if (slice_state) HIP_TRY(hipMemcpy(slice_state, b->d_sstate, 8 * (size_t)b->n_cells * b->S, hipMemcpyDeviceToHost));
  return RS_OK;
}

/* the inverse of rs_batch_read_state for the two arrays a restart needs set: PF averages (RadioBearer::m_averageTransmissionRate,
 * ref: src/flows/radio-bearer.cpp:54 starts them at 100 000) and slice_rbs_offset_ / slice_ewma_time_ */
int rs_batch_write_state(rs_batch* b, const double* avg, const double* slice_state) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  if (b->direct) return fail(RS_ERR_INVALID, "drop-in contexts take the averages per call");
  if (b->queues) return fail(RS_ERR_STATE, "the queue model keeps one average per bearer: not settable through this call");
  const size_t n = (size_t)b->n_cells * b->U;
  if (avg)
    for (size_t i = 0; i < n; i++)
      if (!(avg[i] >= 1.0) || !(avg[i] <= 1e300)) return fail(RS_ERR_INVALID, "avg_rate[%zu] = %g: the EWMA keeps averages at or above 1 (radio-bearer.cpp:160-162)", i, avg[i]);
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (avg) HIP_TRY(hipMemcpy(b->d_avg, avg, 8 * n, hipMemcpyHostToDevice));
  if (slice_state) HIP_TRY(hipMemcpy(b->d_sstate, slice_state, 8 * (size_t)b->n_cells * b->S, hipMemcpyHostToDevice));
  return RS_OK;
}

/* ---- finite queues ---- */
int rs_batch_set_bearers(rs_batch* b, const uint8_t* bearer_kind) {
  if (!b || !bearer_kind) return fail(RS_ERR_INVALID, "null argument");
  if (b->direct) return fail(RS_ERR_INVALID, "drop-in contexts take the queue state per TTI (rs_tti_in.hol_delay / prio_has_data)");
  if (b->ttis_done) return fail(RS_ERR_STATE, "rs_batch_set_bearers after TTIs were run");
  if (b->sched != RS_SCHED_SEQUENTIAL && b->sched != RS_SCHED_MAXCELL && b->sched != RS_SCHED_SUBOPT && b->sched != RS_SCHED_VOGEL &&
      b->sched != RS_SCHED_PF && b->sched != RS_SCHED_NVS)
    return fail(RS_ERR_INVALID, "finite queues: schedulers 1, 7, 8, 9, 101 and 103 only (sched %d)", b->sched);
  const size_t U = b->U;
  for (size_t u = 0; u < U; u++) {
    for (int k = 0; k < 2; k++)
      if (bearer_kind[u * 2 + k] > 2) return fail(RS_ERR_INVALID, "bearer kind %d of user %zu", bearer_kind[u * 2 + k], u);
    if (!bearer_kind[u * 2] && !bearer_kind[u * 2 + 1]) return fail(RS_ERR_INVALID, "user %zu has no bearer", u);
  }
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  /* validate the queue model's LDS carve and build its kernel BEFORE the batch changes: on failure it stays as it was */
  /* Where the bearers' hot words live (RS_QSTATE_BYTES_PER_USER per user).  LDS when they fit the CU's 160 KB -- measured on
   * exp-customize-20slices x 512 cells (profiles/r03_queue_mode.md) -- unless that is what pushes the cell over 80 KB while the
   * batch has more cells than the chip has CUs: then residency would halve the cells per CU, and the words stay in HBM.
   * rs_batch_config.queue_state_lds overrides (1: LDS whenever it fits, -1: HBM). */
  int qmode = 2;
  {
    const RsCarve lds_c = rs_carve(b->S, b->U, b->R, b->sched, b->threads, 2), hbm_c = rs_carve(b->S, b->U, b->R, b->sched, b->threads, 3);
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, b->cfg.cell.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const bool halves = lds_c.q_lds && lds_c.lds_bytes > 80 * 1024 && hbm_c.lds_bytes <= 80 * 1024 && b->n_cells > cus;
    if (b->cfg.queue_state_lds < 0 || (b->cfg.queue_state_lds == 0 && halves)) qmode = 3;
    const RsCarve qc = qmode == 2 ? lds_c : hbm_c;
    if (qc.lds_bytes > 160 * 1024) return fail(RS_ERR_INVALID, "cell needs %d B of LDS with the queue model (> 160 KiB)", qc.lds_bytes);
  }
  RsJitKernel* qjit = nullptr;
  char qmsg[sizeof b->jit_msg] = {0};
  if (b->jit_wanted) { /* the shape-specialised kernel of the queue model is a different code object */
    qjit = rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, b->sched, qmode, slice_window(b), qmsg, sizeof qmsg,
                      b->cfg.cqi_refresh <= 4 ? 2 : 0); /* (the streamed bit as batch_new and lean_kernel set it: the general and the lean build differ in the lean bit only) */
    if (!qjit && !qmsg[0]) snprintf(qmsg, sizeof qmsg, "hiprtc build of the queue-model kernel failed");
    if (!qjit && b->threads > 512) return fail(RS_ERR_INVALID, "threads_per_cell %d needs the shape-specialised queue-model kernel: %s", b->threads, qmsg);
  }
  const size_t n = (size_t)b->n_cells * 2 * U;
  if (!b->d_bearer_kind) {
    HIP_TRY(hipMalloc(&b->d_bearer_kind, 2 * U));
    HIP_TRY(hipMalloc(&b->d_qi, 4 * 7 * n));
    HIP_TRY(hipMalloc(&b->d_bavg, 8 * n));
    HIP_TRY(hipMalloc(&b->d_bcum, 8 * 2 * n));
    HIP_TRY(hipMalloc(&b->d_qflags, (size_t)b->n_cells * U));
    HIP_TRY(hipMalloc(&b->d_qhol, 8 * (size_t)b->n_cells * U));
  }
  HIP_TRY(hipMemcpy(b->d_bearer_kind, bearer_kind, 2 * U, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(b->d_qi, 0, 4 * 7 * n));
  HIP_TRY(hipMemset(b->d_bcum, 0, 8 * 2 * n));
  HIP_TRY(hipMemset(b->d_qflags, 0, (size_t)b->n_cells * U));
  HIP_TRY(hipMemset(b->d_qhol, 0, 8 * (size_t)b->n_cells * U));
  std::vector<double> avg(n, 100000.0); /* radio-bearer.cpp:54 */
  HIP_TRY(hipMemcpy(b->d_bavg, avg.data(), 8 * n, hipMemcpyHostToDevice));
  /* commit: mode, carve and kernel together (the queue = 0 kernel must never run with queue arguments) */
  b->queues = true;
  b->qmode = qmode;
  carve_lds(b, &b->base); /* schedulers 1 and 7 keep per-bearer scratch in LDS in this mode */
  if (b->jit_wanted) {
    b->jit = qjit; /* nullptr: the built-in queue kernels run, rs_batch_jit_status says why */
    b->jit_lean = nullptr; /* the lean build belongs to the kernel it is a build of: the queue model's is compiled on demand */
    b->jit_lean_tried = false;
    snprintf(b->jit_msg, sizeof b->jit_msg, "%s", qmsg);
  }
  return RS_OK;
}

int rs_batch_set_arrivals(rs_batch* b, const int64_t* offsets, const double* time, const int32_t* n_full, const int32_t* last_bytes) {
  if (!b || !offsets) return fail(RS_ERR_INVALID, "null argument");
  if (!b->queues) return fail(RS_ERR_STATE, "rs_batch_set_bearers first");
  if (b->ttis_done) return fail(RS_ERR_STATE, "rs_batch_set_arrivals after TTIs were run");
  const size_t nb = (size_t)b->n_cells * b->U * 2;
  if (offsets[0] != 0) return fail(RS_ERR_INVALID, "offsets[0] must be 0");
  for (size_t i = 0; i < nb; i++) {
    if (offsets[i + 1] < offsets[i]) return fail(RS_ERR_INVALID, "offsets must not decrease (bearer %zu)", i);
    for (int64_t k = offsets[i]; k < offsets[i + 1]; k++) {
      if (k > offsets[i] && !(time[k] >= time[k - 1])) return fail(RS_ERR_INVALID, "arrival times of bearer %zu are not ascending", i);
      if (n_full[k] < 0 || last_bytes[k] < 0 || last_bytes[k] > RS_FULL_PACKET || n_full[k] + last_bytes[k] == 0)
        return fail(RS_ERR_INVALID, "burst %lld of bearer %zu: n_full %d, last %d", (long long)k, i, n_full[k], last_bytes[k]);
    }
  }
  const size_t total = (size_t)offsets[nb];
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  void* old[] = {b->d_arr_off, b->d_arr_time, b->d_arr_nfull, b->d_arr_last};
  for (void* q : old)
    if (q) HIP_TRY(hipFree(q));
  b->d_arr_off = nullptr; b->d_arr_time = nullptr; b->d_arr_nfull = nullptr; b->d_arr_last = nullptr;
  HIP_TRY(hipMalloc(&b->d_arr_off, 8 * (nb + 1)));
  HIP_TRY(hipMalloc(&b->d_arr_time, 8 * (total ? total : 1)));
  HIP_TRY(hipMalloc(&b->d_arr_nfull, 4 * (total ? total : 1)));
  HIP_TRY(hipMalloc(&b->d_arr_last, 4 * (total ? total : 1)));
  HIP_TRY(hipMemcpy(b->d_arr_off, offsets, 8 * (nb + 1), hipMemcpyHostToDevice));
  if (total) {
    HIP_TRY(hipMemcpy(b->d_arr_time, time, 8 * total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_arr_nfull, n_full, 4 * total, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->d_arr_last, last_bytes, 4 * total, hipMemcpyHostToDevice));
  }
  return RS_OK;
}

int rs_batch_read_bearer_state(rs_batch* b, double* avg, int64_t* cum_bytes, int64_t* cum_rbs, int32_t* queue_bytes,
                               int32_t* queue_packets) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  if (!b->queues) return fail(RS_ERR_STATE, "the batch has no bearers (rs_batch_set_bearers)");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const size_t U = b->U, n = (size_t)b->n_cells * 2 * U;
  /* device layout [cells][2][U] -> [cells][U][2] */
  auto gather = [&](auto* dst, const auto* src_dev, auto zero) -> int {
    std::vector<decltype(zero)> tmp(n);
    HIP_TRY(hipMemcpy(tmp.data(), src_dev, sizeof(zero) * n, hipMemcpyDeviceToHost));
    for (size_t c = 0; c < (size_t)b->n_cells; c++)
      for (size_t k = 0; k < 2; k++)
        for (size_t u = 0; u < U; u++) dst[(c * U + u) * 2 + k] = tmp[(c * 2 + k) * U + u];
    return RS_OK;
  };
  int rc = RS_OK;
  if (avg && !rc) rc = gather(avg, b->d_bavg, 0.0);
  if (cum_bytes && !rc) rc = gather(cum_bytes, b->d_bcum, (int64_t)0);
  if (cum_rbs && !rc) rc = gather(cum_rbs, b->d_bcum + n, (int64_t)0);
  if (queue_bytes && !rc) rc = gather(queue_bytes, b->d_qi + 4 * n, (int32_t)0);
  if (queue_packets && !rc) rc = gather(queue_packets, b->d_qi + 5 * n, (int32_t)0);
  return rc;
}

/* InternetFlow's arrival process (ref: src/flows/application/InternetFlow.cpp:41-58, 83-90, 155-200): exponential
 * inter-arrival times from libstdc++'s std::exponential_distribution over a default-constructed std::default_random_engine
 * (every InternetFlow object owns one, so every flow of a rate sees the same intervals), rounded up to whole
 * milliseconds; heavy-tailed flow sizes from the table's CDF with one rand() per flow.  The reference draws the sizes from the
 * process-wide libc stream it shares with the scheduler; a batch has no such global event order, so each bearer gets its
 * own libc-compatible stream seeded with size_seed (documented deviation of the batched traffic model). */
int rs_internet_flow_arrivals(double rate_mbps, double start_time, double stop_time, uint32_t size_seed, int32_t max_bursts,
                              double* time, int32_t* n_full, int32_t* last_bytes) {
  if (!(rate_mbps > 0) || !(stop_time > start_time) || max_bursts < 1 || !time || !n_full || !last_bytes)
    return fail(RS_ERR_INVALID, "bad argument");
  static const int kFlowSize[11] = {1460, 2920, 4380, 7300, 10220, 58400, 105120, 200020, 389820, 1733020, 3076220};
  static const double kFlowCdf[11] = {0.5, 0.6, 0.7, 0.75, 0.8, 0.8125, 0.825, 0.85, 0.9, 0.95, 1};
  double avg_flowsize = 0;
  for (int i = 0; i < 11; i++) avg_flowsize += (i == 0 ? kFlowCdf[0] : kFlowCdf[i] - kFlowCdf[i - 1]) * kFlowSize[i];
  const int avg_size = (int)avg_flowsize;
  const double interval_mean = avg_size / rate_mbps * 8 / 1000000;
  const double lambda = 1 / interval_mean;
  std::exponential_distribution<double> distribute(lambda);
  std::default_random_engine generator;
  HostRng sizes;
  sizes.seed(size_seed);
  double now = start_time; /* DoStart schedules the first Send at +0.0 from the application's start */
  int n = 0;
  for (;;) {
    /* Send(): one flow */
    const double cdf = (double)sizes.next() / 2147483647; /* rand() / RAND_MAX */
    int flow_size = kFlowSize[10];
    for (int i = 0; i < 11; i++)
      if (kFlowCdf[i] >= cdf) { flow_size = kFlowSize[i]; break; }
    const int last_pkt = flow_size % 1490;
    const int n_pkts = (int)std::ceil(flow_size / (double)1490);
    if (n >= max_bursts) return fail(RS_ERR_INVALID, "more than max_bursts = %d flows before stop_time", max_bursts);
    time[n] = now;
    /* every packet gets its headers, then the last one is SET to the remainder (no headers on top): :133-136 */
    n_full[n] = last_pkt != 0 ? n_pkts - 1 : n_pkts;
    last_bytes[n] = last_pkt;
    n++;
    /* ScheduleTransmit(GetInterval()) */
    const double interval = std::ceil(distribute(generator) * 1000) / 1000.0;
    if (!((now + interval) < stop_time)) break;
    now = interval + now; /* Simulator::DoSchedule: timeStamp = time + Now() */
  }
  return n;
}

int rs_batch_read_clock(rs_batch* b, double* t, double* last_update) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::vector<RsCellScalars> sc(b->n_cells);
  HIP_TRY(hipMemcpy(sc.data(), b->d_scal, sizeof(RsCellScalars) * b->n_cells, hipMemcpyDeviceToHost));
  for (int c = 0; c < b->n_cells; c++) {
    if (t) t[c] = sc[c].t;
    if (last_update) last_update[c] = sc[c].last_update;
  }
  return RS_OK;
}

int rs_batch_debug_heap_sorts(rs_batch* b, int64_t* out) {
  if (!b || !out) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::vector<RsCellScalars> sc(b->n_cells);
  HIP_TRY(hipMemcpy(sc.data(), b->d_scal, sizeof(RsCellScalars) * b->n_cells, hipMemcpyDeviceToHost));
  for (int c = 0; c < b->n_cells; c++)
    for (int k = 0; k < 3; k++) out[c * 3 + k] = sc[c].heap_sorts[k];
  return RS_OK;
}

/* ---- checkpoint / resume (ABI 10): everything a batch carries from one launch to the next, as one host block ----
 * header: magic "RSCK2", the shape it belongs to (S, U, R, G, sched, cells, queue model, threads are NOT part of it: any workgroup
 * size continues a run), the number of scheduled TTIs so far, the kernel family that wrote the pending-grant words; then the device
 * arrays verbatim.  The CQI source (epoch grids / trace tables / arrival bursts) is configuration, not state: the caller sets it
 * again on the batch that resumes. */
namespace {
struct CkptHeader {
  char magic[8];
  int32_t S, U, R, G, sched, n_cells, queues, packed_tx; /* packed_tx: the pending-grant words were left by a shape-specialised kernel */
  int64_t ttis_done;
  uint64_t bytes; /* of the whole block */
  /* round 6: FNV-1a over everything else that decides how the run continues -- the users' slices, the slices' weights and
   * algorithm parameters, cqi_refresh, first_tti, phy_error_draws, synthetic_exp -- and over the layout of the block itself (ABI
   * version, sizeof(RsCellScalars)): a checkpoint of a batch whose results would differ is refused, not continued silently */
  uint64_t config_hash;
};
uint64_t ckpt_config_hash(const rs_batch* b) {
  unsigned long long h = 1469598103934665603ull;
  auto mix = [&h](const void* q, size_t n) {
    const unsigned char* c = (const unsigned char*)q;
    for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; }
    h ^= 0xffu; h *= 1099511628211ull; /* field boundary */
  };
  const int32_t scalars[6] = {RS_ABI_VERSION, (int32_t)sizeof(RsCellScalars), b->cfg.cqi_refresh, b->cfg.first_tti, b->cfg.phy_error_draws ? 1 : 0,
                              b->synthetic ? 1 : 0};
  mix(scalars, sizeof scalars);
  mix(b->u2s.data(), 4 * b->u2s.size());
  mix(b->weight.data(), 8 * b->weight.size());
  mix(b->alpha.data(), 4 * b->alpha.size());
  mix(b->beta.data(), 4 * b->beta.size());
  mix(b->eps.data(), 4 * b->eps.size());
  mix(b->psi.data(), 4 * b->psi.size());
  return h;
}
}  // namespace

int64_t rs_batch_checkpoint_bytes(rs_batch* b) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  if (b->direct) return fail(RS_ERR_INVALID, "drop-in contexts carry slice_rbs_offset_ only: rs_get_slice_offset / rs_set_slice_offset");
  size_t n = sizeof(CkptHeader);
  for (const StatePart& q : state_parts(b)) n += q.n;
  return (int64_t)n;
}

int rs_batch_checkpoint_save(rs_batch* b, void* buf, size_t buflen) {
  const int64_t need = rs_batch_checkpoint_bytes(b);
  if (need < 0) return (int)need;
  if (!buf || buflen < (size_t)need) return fail(RS_ERR_INVALID, "checkpoint buffer of %zu bytes, %lld needed", buflen, (long long)need);
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  CkptHeader h{};
  memcpy(h.magic, "RSCK2\0\0", 8);
  h.S = b->S; h.U = b->U; h.R = b->R; h.G = b->G; h.sched = b->sched; h.n_cells = b->n_cells; h.queues = b->queues ? 1 : 0;
  h.packed_tx = (b->jit && !b->queues) ? 1 : 0;
  h.ttis_done = b->ttis_done;
  h.bytes = (uint64_t)need;
  h.config_hash = ckpt_config_hash(b);
  memcpy(buf, &h, sizeof h);
  char* out = (char*)buf + sizeof h;
  for (const StatePart& q : state_parts(b)) {
    HIP_TRY(hipMemcpy(out, q.p, q.n, hipMemcpyDeviceToHost));
    out += q.n;
  }
  return RS_OK;
}

int rs_batch_checkpoint_load(rs_batch* b, const void* buf, size_t buflen) {
  const int64_t need = rs_batch_checkpoint_bytes(b);
  if (need < 0) return (int)need;
  if (!buf || buflen < sizeof(CkptHeader)) return fail(RS_ERR_INVALID, "no checkpoint");
  CkptHeader h;
  memcpy(&h, buf, sizeof h);
  if (memcmp(h.magic, "RSCK2\0\0", 8) != 0) return fail(RS_ERR_INVALID, "not a checkpoint of this library version (magic)");
  if (h.S != b->S || h.U != b->U || h.R != b->R || h.G != b->G || h.sched != b->sched || h.n_cells != b->n_cells || h.queues != (b->queues ? 1 : 0))
    return fail(RS_ERR_INVALID, "checkpoint of another batch: %d slices x %d UEs x %d RBGs x %d PRBs, sched %d, %d cells, queue model %d", h.S, h.U, h.R,
                h.G, h.sched, h.n_cells, h.queues);
  if (h.bytes != (uint64_t)need || buflen < (size_t)need) return fail(RS_ERR_INVALID, "checkpoint of %llu bytes, this batch's are %lld", (unsigned long long)h.bytes, (long long)need);
  if (h.config_hash != ckpt_config_hash(b))
    return fail(RS_ERR_INVALID, "checkpoint of a batch configured differently (users per slice, slice weights / algorithm parameters, cqi_refresh, first_tti, "
                                "phy_error_draws, synthetic_exp or the library's state layout): the run would not continue as it began");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  const char* in = (const char*)buf + sizeof h;
  for (const StatePart& q : state_parts(b)) {
    if (q.p == (void*)b->d_tx && !b->queues) {
      /* the pending-grant words follow the kernel family that will consume them: a shape-specialised kernel wants bytes | PRBs << 20 with
       * "counted" set (the grant is in the cumulative totals already, whoever wrote it), the built-in kernels want plain bytes */
      std::vector<int32_t> tx(q.n / 4);
      memcpy(tx.data(), in, q.n);
      const bool want_packed = b->jit != nullptr;
      for (int32_t& w : tx) {
        if (want_packed && !h.packed_tx) w = w ? (w | RS_TX_COUNTED) : 0; /* (no PRB count: it is in the totals, nothing reads it again) */
        else if (!want_packed && h.packed_tx) w &= RS_TX_BYTES_MASK;
      }
      HIP_TRY(hipMemcpy(q.p, tx.data(), q.n, hipMemcpyHostToDevice));
    } else {
      HIP_TRY(hipMemcpy(q.p, in, q.n, hipMemcpyHostToDevice));
    }
    in += q.n;
  }
  b->ttis_done = h.ttis_done;
  return RS_OK;
}

int rs_batch_debug_clocks(rs_batch* b, double* shader_mhz, double* kernel_ms) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  std::vector<RsCellScalars> sc(b->n_cells);
  HIP_TRY(hipMemcpy(sc.data(), b->d_scal, sizeof(RsCellScalars) * b->n_cells, hipMemcpyDeviceToHost));
  for (int c = 0; c < b->n_cells; c++) {
    const double real = (double)(sc[c].real_end - sc[c].real_begin), clk = (double)(sc[c].clk_end - sc[c].clk_begin);
    if (shader_mhz) shader_mhz[c] = real > 0 ? clk / real * 100.0 : 0.0; /* s_memrealtime ticks at 100 MHz */
    if (kernel_ms) kernel_ms[c] = real / 1e5;
  }
  return RS_OK;
}

int rs_batch_autotune_report(rs_batch* b, char* msg, size_t msglen) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  if (msg && msglen) snprintf(msg, msglen, "%s", b->autotune_msg);
  return b->autotune_n;
}

int rs_batch_prepare_launch(rs_batch* b, int32_t n_ttis) {
  if (!b || n_ttis < 1) return fail(RS_ERR_INVALID, "bad argument");
  if (b->cqi_mode == RS_CQI_NONE) return fail(RS_ERR_STATE, "no CQI source set");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  (void)lean_kernel(b, n_ttis > RS_MAX_TTIS_PER_LAUNCH ? RS_MAX_TTIS_PER_LAUNCH : n_ttis, false);
  if (!b->selfchecked) {
    const int rc = selfcheck(b, n_ttis > RS_MAX_TTIS_PER_LAUNCH ? RS_MAX_TTIS_PER_LAUNCH : n_ttis, false);
    if (rc) return rc;
  }
  if (b->cfg.autotune && !b->autotuned) {
    const int rc = autotune(b, n_ttis > RS_MAX_TTIS_PER_LAUNCH ? RS_MAX_TTIS_PER_LAUNCH : n_ttis);
    if (rc) return rc;
  }
  return RS_OK;
}

int rs_batch_jit_status(rs_batch* b, char* msg, size_t msglen) {
  if (!b) return fail(RS_ERR_INVALID, "null batch");
  if (msg && msglen) snprintf(msg, msglen, "%s", b->jit_msg);
  if (b->jit_rejected) return -2;
  if (msg && msglen && b->jit && !b->jit_msg[0] && b->selfcheck_msg[0]) snprintf(msg, msglen, "%s", b->selfcheck_msg);
  return b->jit ? 1 : (b->jit_wanted ? -1 : 0);
}

int rs_batch_slice_bytes_device(rs_batch* b, uint64_t* d_out) {
  if (!b || !d_out) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  HIP_TRY(rs_launch_slice_bytes(b->d_cumb, b->d_user_slice, b->n_cells, b->U, b->S, (unsigned long long*)d_out, b->stream));
  return RS_OK;
}

int rs_batch_slice_bytes(rs_batch* b, uint64_t* h_out) {
  if (!b || !h_out) return fail(RS_ERR_INVALID, "null argument");
  int rc = rs_batch_slice_bytes_device(b, (uint64_t*)b->d_slice_bytes);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(h_out, b->d_slice_bytes, 8 * b->S, hipMemcpyDeviceToHost));
  return RS_OK;
}

int rs_batch_debug_stamps(rs_batch* b, int32_t cell, uint64_t* out20) {
  if (!b || !out20 || cell < 0 || cell >= b->n_cells) return fail(RS_ERR_INVALID, "bad argument");
  if (!b->d_stamps) return fail(RS_ERR_STATE, "not a diagnostic (-DRS_STAMPS) build");
  HIP_TRY(hipStreamSynchronize(b->stream));
  HIP_TRY(hipMemcpy(out20, b->d_stamps + (size_t)cell * 20, 8 * 20, hipMemcpyDeviceToHost));
  return RS_OK;
}

int64_t rs_batch_ttis_done(rs_batch* b) { return b ? b->ttis_done : -1; }
void* rs_batch_stream(rs_batch* b) { return b ? (void*)b->stream : nullptr; }
const char* rs_batch_kernel_name(rs_batch* b) {
  if (!b) return "";
  if (b->jit) return "rs_cell_kernel_jit";
  switch (b->sched) {
    case 1: return "rs_cell_kernel<1, 0, false>";
    case 7: return "rs_cell_kernel<7, 0, false>";
    case 8: return "rs_cell_kernel<8, 0, false>";
    case RS_SCHED_SUBOPT: return "rs_cell_kernel<101, 0, false>";
    case RS_SCHED_VOGEL: return "rs_cell_kernel<103, 0, false>";
    case RS_SCHED_NVS_NONGREEDY: return "rs_cell_kernel<11, 0, false>";
    case RS_SCHED_UPPERBOUND: {
      const int ept = (b->R * b->S + b->threads - 1) / b->threads;
      return ept <= 1 ? "rs_cell_kernel<10, 1, false>" : ept <= 2 ? "rs_cell_kernel<10, 2, false>" : ept <= 3 ? "rs_cell_kernel<10, 3, false>" : "rs_cell_kernel<10, 4, false>";
    }
    default: {
      const int ept = (b->R * b->S + b->threads - 1) / b->threads;
      return ept <= 1 ? "rs_cell_kernel<9, 1, false>" : ept <= 2 ? "rs_cell_kernel<9, 2, false>" : ept <= 3 ? "rs_cell_kernel<9, 3, false>" : ept <= 4 ? "rs_cell_kernel<9, 4, false>" : "rs_cell_kernel<9, 0, false>";
    }
  }
}

}  // extern "C"

/* ===================================================================================== */
/* drop-in mode: one RBsAllocation() per call on a one-cell batch                        */

struct rs_ctx {
  rs_batch* b = nullptr;
  /* one pinned staging block each way, mirrored by one device block each way:
   *   in : [ CQI grid, stride bytes | user slice ids, round16(U) | avg_rate f64[U] ]
   *   out: [ tbs i32[U] | uinfo i32[U] | map i16[R] | quota i16[S] | target i16[S] ] */
  uint8_t *h_in = nullptr, *d_in = nullptr, *h_out = nullptr, *d_out = nullptr;
  size_t in_bytes = 0, out_bytes = 0;
  /* zero-copy: the kernel reads the pinned input block and writes the pinned output block itself (a few KB over the host
   * link inside one launch) instead of two hipMemcpyAsync around it; RS_DROPIN_COPY=1 restores the copies */
  uint8_t *z_in = nullptr, *z_out = nullptr; /* device-side addresses of h_in / h_out; null: copy path */
  /* RS_DROPIN_TIMING=1: host-side breakdown of the calls (prepare / enqueue / wait / unpack), printed by rs_destroy.
   * (Polling hipStreamQuery instead of hipStreamSynchronize was measured: no faster, the runtime already waits actively.) */
  bool timing = false;
  double t_prep = 0, t_enq = 0, t_wait = 0, t_unpack = 0;
  long n_calls = 0;
  /* Completion by a word in the pinned output block (round 5): the kernel's last store writes the call's sequence number there and
   * the host spins on it -- the stream's completion signal (interrupt or the runtime's own polling, then its bookkeeping) costs a
   * few microseconds more per call (profiles/r05_dropin_split.md).  Zero-copy calls only; RS_DROPIN_POLL=0 keeps
   * hipStreamSynchronize; a call that does not see its number within RS_DROPIN_POLL_US (default 2 000) microseconds falls back to
   * it (and reports whatever error the stream holds). */
  bool poll = false;
  uint32_t seq = 0;
  size_t flag_off = 0; /* offset of the word in h_out / z_out: behind the largest layout, on its own cache line */
  long poll_us = 2000;
  long n_polled = 0, n_fallback = 0;
  /* rs_tti_in.cqi_epoch (round 6): the grid of the last call whose reports were new, kept on the device as the LDS image ([R][Upad]);
   * a call with the same epoch number and the same user list reads it instead of the caller's block (RsLaunch::image_mode) */
  uint8_t* d_img = nullptr;
  bool img_valid = false, img_prb = false, img_has_ids = false;
  uint64_t img_epoch = 0;
  int img_n = 0;
  std::vector<int32_t> img_ids;
  long n_img_reused = 0;
  /* the specialised builds' check against the built-in kernel (rs_ctx_specialize; index 0: general build, 1: lean build):
   * calls still to be checked, calls that agreed so far */
  int chk_left[2] = {0, 0}, chk_agreed[2] = {0, 0};
  bool jit_dropped = false;
  uint8_t *d_out2 = nullptr, *d_chk = nullptr; /* the built-in kernel's output block; slice state + scalars before / after it */
  std::vector<uint8_t> h_out2;
  char jit_msg[512] = "";
};

namespace {
/* drop-in contexts alive in this process: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
 * environment says otherwise), and one-TTI kernels of contexts that share a hardware queue run one after the other
 * (profiles/r06_dropin_concurrency.md: nine threads, four queues, at most four kernels in flight) */
std::atomic<int> g_live_ctx{0};
struct CtxLayout {
  size_t grid, slice, avg, hol, prio, gate, draws, prb, in_total, tbs, uinfo, map, quota, target, upper, out_total;
};
/* first CQI outside 1..15, or null: a branch-free pass (vectorised by the compiler: 12.5 KB per call at 500 UEs x 25 RBGs),
 * the offender is only looked for when there is one */
const uint8_t* first_bad_cqi(const uint8_t* q, size_t n) {
  unsigned bad = 0;
  for (size_t i = 0; i < n; i++) bad |= (unsigned)((uint8_t)(q[i] - 1) > 14);
  if (!bad) return nullptr;
  for (size_t i = 0; i < n; i++)
    if (q[i] < 1 || q[i] > 15) return q + i;
  return nullptr;
}
CtxLayout ctx_layout(int n, int R, int S, int G, bool with_draws) {
  CtxLayout l;
  l.grid = 0;
  l.slice = round_up(n * R, 16);
  l.avg = l.slice + round_up(n, 16);
  l.hol = l.avg + 8 * (size_t)n;
  l.prio = l.hol + 8 * (size_t)n;
  l.gate = l.prio + round_up(n, 16);
  l.draws = l.gate + round_up(4 * n, 16);
  l.prb = l.draws + (with_draws ? round_up(RS_NVS_SAMPLES * n, 16) : 0);
  l.in_total = l.prb + round_up(n * R * G, 16);
  l.tbs = 0;
  l.uinfo = 4 * (size_t)n;
  l.map = 8 * (size_t)n;
  l.quota = l.map + round_up(2 * R, 8);
  l.target = l.quota + round_up(2 * S, 8);
  l.upper = l.target + round_up(2 * S, 8);
  l.out_total = l.upper + 4 * (size_t)S * R; /* sched 10 only uses it */
  return l;
}
}  // namespace

extern "C" {

int rs_ctx_debug_heap_sorts(rs_ctx* ctx, int64_t* out) {
  if (!ctx || !ctx->b) return fail(RS_ERR_INVALID, "null context");
  return rs_batch_debug_heap_sorts(ctx->b, out);
}

rs_ctx* rs_create_checked(const rs_config* cfg, int abi_version, size_t cfg_size) {
  if (abi_version != RS_ABI_VERSION || cfg_size != sizeof(rs_config)) {
    fail(RS_ERR_INVALID, "ABI mismatch: caller was built against ABI %d with an rs_config of %zu bytes, this library is ABI %d with %zu bytes",
         abi_version, cfg_size, RS_ABI_VERSION, sizeof(rs_config));
    return nullptr;
  }
  return rs_create(cfg);
}

rs_ctx* rs_create(const rs_config* cfg) {
  rs_batch_config bc;
  memset(&bc, 0, sizeof bc);
  if (!cfg) { fail(RS_ERR_INVALID, "null config"); return nullptr; }
  bc.cell = *cfg;
  bc.n_cells = 1;
  bc.cqi_refresh = 1;
  rs_batch* b = batch_new(&bc, true);
  if (!b) return nullptr;
  rs_ctx* c = new (std::nothrow) rs_ctx();
  if (!c) { rs_batch_destroy(b); fail(RS_ERR_INVALID, "out of memory"); return nullptr; }
  c->b = b;
  const CtxLayout l = ctx_layout(b->U, b->R, b->S, b->G, b->sched == RS_SCHED_NVS_NONGREEDY);
  c->in_bytes = l.in_total;
  c->flag_off = round_up(l.out_total, 64);
  c->out_bytes = c->flag_off + 64;
  const size_t img_bytes = round_up(rs_upad_of(b->U) * b->R, 16);
  const size_t chk_bytes = round_up(8 * b->S, 256) + round_up((int)sizeof(RsCellScalars), 256);
  bool ok = hipMalloc(&c->d_in, c->in_bytes) == hipSuccess && hipMalloc(&c->d_out, c->out_bytes) == hipSuccess &&
            hipMalloc(&c->d_img, img_bytes) == hipSuccess && hipMalloc(&c->d_out2, c->out_bytes) == hipSuccess &&
            hipMalloc(&c->d_chk, 2 * chk_bytes) == hipSuccess &&
            hipHostMalloc((void**)&c->h_in, c->in_bytes, hipHostMallocMapped) == hipSuccess &&
            hipHostMalloc((void**)&c->h_out, c->out_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
  const char* force_copy = getenv("RS_DROPIN_COPY");
  if (ok && !(force_copy && force_copy[0] == '1')) {
    void *zi = nullptr, *zo = nullptr;
    if (hipHostGetDevicePointer(&zi, c->h_in, 0) == hipSuccess && hipHostGetDevicePointer(&zo, c->h_out, 0) == hipSuccess) {
      c->z_in = (uint8_t*)zi;
      c->z_out = (uint8_t*)zo;
    }
  }
  if (!ok) { fail(RS_ERR_HIP, "allocation of the staging blocks failed"); rs_destroy(c); return nullptr; }
  const char* tm = getenv("RS_DROPIN_TIMING");
  c->timing = tm && tm[0] == '1';
  const char* pl = getenv("RS_DROPIN_POLL");
  c->poll = c->z_out != nullptr && !(pl && pl[0] == '0');
  if (const char* pu = getenv("RS_DROPIN_POLL_US")) c->poll_us = atol(pu) > 0 ? atol(pu) : 2000;
  if (c->h_out) memset(c->h_out + c->flag_off, 0, 64);
  c->h_out2.assign(c->out_bytes, 0);
  g_err[0] = 0;
  if (cfg->link_tables == RS_LINK_DEFAULT) {
    /* a drop-in context follows THIS host's libm (the reference beside it does); when that is not the libm of the fixtures, say so once */
    char where[300] = "";
    const int diff = rs_link_tables_compare(where, sizeof where);
    if (diff > 0)
      snprintf(g_err, sizeof g_err, "warning: this host's libm gives %d EESM constant(s) that differ from the pinned glibc-2.35 set (%.300s): the context "
               "follows the host (RS_LINK_HOST_LIBM); RS_LINK_PINNED_GLIBC_2_35 reproduces the fixtures", diff, where);
  }
  {
    const int live = ++g_live_ctx;
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int hwq = e && atoi(e) > 0 ? atoi(e) : 4;
    if (live > hwq && !cfg->stream) {
      const size_t at = strlen(g_err);
      snprintf(g_err + at, sizeof g_err - at, "%swarning: %d drop-in contexts in this process share %d hardware queues (GPU_MAX_HW_QUEUES, read by the HIP "
               "runtime at its first call): their one-TTI kernels run %d at a time; export GPU_MAX_HW_QUEUES=%d (or more) before the process starts",
               at ? "; " : "", live, hwq, hwq, live);
    }
  }
  return c;
}

void rs_destroy(rs_ctx* c) {
  if (!c) return;
  if (c->h_out2.size()) --g_live_ctx; /* (counted by rs_create once it got that far) */
  if (c->timing && c->n_calls)
    fprintf(stderr, "rs_schedule_tti x %ld: prepare %.2f us, enqueue %.2f us, wait %.2f us, unpack %.2f us per call (%ld completed by the polled word, %ld fell back to the stream, %ld read the device-resident CQI image)\n", c->n_calls,
            c->t_prep / c->n_calls, c->t_enq / c->n_calls, c->t_wait / c->n_calls, c->t_unpack / c->n_calls, c->n_polled, c->n_fallback, c->n_img_reused);
  if (c->b && c->b->stream) (void)hipStreamSynchronize(c->b->stream);
  if (c->d_in) (void)hipFree(c->d_in);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_img) (void)hipFree(c->d_img);
  if (c->d_out2) (void)hipFree(c->d_out2);
  if (c->d_chk) (void)hipFree(c->d_chk);
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_out) (void)hipHostFree(c->h_out);
  rs_batch_destroy(c->b);
  delete c;
}

int rs_schedule_tti(rs_ctx* c, const rs_tti_in* in, rs_tti_out* out) {
  if (!c || !in || !out) return fail(RS_ERR_INVALID, "null argument");
  rs_batch* b = c->b;
  using clk = std::chrono::steady_clock;
  const clk::time_point t0 = c->timing ? clk::now() : clk::time_point();
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  const int n = in->n_users, R = b->R, S = b->S;
  if (n < 1 || n > b->U) return fail(RS_ERR_INVALID, "n_users %d outside 1..%d", n, b->U);
  if ((!in->cqi && !in->cqi_prb) || !in->avg_rate) return fail(RS_ERR_INVALID, "null cqi/avg_rate");
  if (!out->rbg_to_user || !out->user_tbs_bits) return fail(RS_ERR_INVALID, "null output array");
  const CtxLayout l = ctx_layout(n, R, S, b->G, b->sched == RS_SCHED_NVS_NONGREEDY);
  uint8_t* h_slice = c->h_in + l.slice;
  for (int i = 0; i < n; i++) {
    int id = in->user_id ? in->user_id[i] : i;
    if (id < 0 || id >= b->U) return fail(RS_ERR_INVALID, "user id %d out of range", id);
    if (i && in->user_id && in->user_id[i] <= in->user_id[i - 1]) return fail(RS_ERR_INVALID, "user_id must ascend");
    h_slice[i] = (uint8_t)b->u2s[id];
    if ((b->sched == RS_SCHED_NVS || b->sched == RS_SCHED_NVS_NONGREEDY) && h_slice[i] != h_slice[0])
      return fail(RS_ERR_INVALID, "RS_SCHED_NVS*: pass only the users of the served slice");
  }
  const int G = b->G;
  size_t in_bytes = l.prb; /* the per-PRB block travels only when given */
  /* rs_tti_in.cqi_epoch: the same reports for the same users as the call before?  Then the caller's block is not touched -- the kernel
   * reads the image the context kept on the device (and, per-PRB reports, the device copy of the block) */
  const bool same_users = c->img_valid && c->img_n == n && c->img_has_ids == (in->user_id != nullptr) &&
                          (!in->user_id || memcmp(c->img_ids.data(), in->user_id, 4 * (size_t)n) == 0);
  const bool reuse_grid = in->cqi_epoch != 0 && same_users && in->cqi_epoch == c->img_epoch && c->img_prb == (in->cqi_prb != nullptr);
  const int image_mode = in->cqi_epoch == 0 ? 0 : (reuse_grid ? 2 : 1);
  if (in->cqi_prb) {
    const size_t np = (size_t)n * R * G;
    if (!reuse_grid) {
      if (const uint8_t* bad = first_bad_cqi(in->cqi_prb, np)) return fail(RS_ERR_INVALID, "CQI %d outside 1..15", *bad);
      memcpy(c->h_in + l.prb, in->cqi_prb, np);
      for (int i = 0; i < n; i++)
        for (int r = 0; r < R; r++) c->h_in[l.grid + (size_t)i * R + r] = in->cqi_prb[((size_t)i * R + r) * G];
    }
    in_bytes = l.prb + np;
  } else if (!reuse_grid) {
    if (const uint8_t* bad = first_bad_cqi(in->cqi, (size_t)n * R)) return fail(RS_ERR_INVALID, "CQI %d outside 1..15", *bad);
    memcpy(c->h_in + l.grid, in->cqi, (size_t)n * R);
  }
  if (!reuse_grid) memset(c->h_in + l.grid + (size_t)n * R, 0, l.slice - (size_t)n * R);
  c->img_valid = false; /* (until this call has gone through) */
  memcpy(c->h_in + l.avg, in->avg_rate, 8 * (size_t)n);
  if (b->gen_exp) {
    /* general exponents: pow(averageRate / 1000.0, psi) with averageRate = 1 + the caller's sum (ref: :681-693), host libm */
    double* den = (double*)(c->h_in + l.avg);
    for (int i = 0; i < n; i++) {
      double k = 1;
      k += in->avg_rate[i];
      k /= 1000.0;
      den[i] = pow(k, b->psi[h_slice[i]]);
    }
  }
  /* The metric scan ranks users with an FP32 product first; its error bound needs every factor to be an
   * ordinary FP32 number.  The reference takes any double (downlink-transport-scheduler.cpp:677-713: a non-finite, huge, tiny
   * or negative average simply flows through the division and the strict '>' scan), so inputs outside the safe range switch
   * this call to the exact FP64 scan of every user instead of being rejected.  (The EWMA keeps real averages in [1, ~1e12].) */
  bool exact_scan = b->gen_exp; /* (powers of any size: no FP32 ranking, every user is compared with the reference's expression) */
  if (!exact_scan) {
    auto ordinary = [](double x, double lo, double hi) { return x >= lo && x <= hi; }; /* false for NaN */
    for (int i = 0; i < n; i++) {
      const double a = in->avg_rate[i];
      const double k = b->sched == RS_SCHED_PF ? a : (1 + a) / 1000.0;
      exact_scan |= !ordinary(k, 0x1p-60, 0x1p60);
    }
    if (b->any_alpha && in->hol_delay)
      for (int i = 0; i < n; i++) {
        const double h = in->hol_delay[i];
        exact_scan |= !(h == 0 || ordinary(h, 0x1p-40, 0x1p40));
      }
  }
  if (b->sched == RS_SCHED_NVS_NONGREEDY) {
    if (!in->rand_draws) return fail(RS_ERR_INVALID, "RS_SCHED_NVS_NONGREEDY needs rand_draws (%d x n_users values)", RS_NVS_SAMPLES);
    const size_t nd = (size_t)RS_NVS_SAMPLES * n;
    for (size_t i = 0; i < nd; i++) {
      if (in->rand_draws[i] < 0) return fail(RS_ERR_INVALID, "rand_draws[%zu] = %d is not a rand() value", i, in->rand_draws[i]);
      c->h_in[l.draws + i] = (uint8_t)(in->rand_draws[i] % 4); /* downlink-nvs-scheduler.cpp:438 */
    }
  }
  const int32_t* gate = b->sched == RS_SCHED_NVS ? in->required_rbs : (b->sched == RS_SCHED_PF ? in->data_to_transmit : nullptr);
  if (gate) {
    for (int i = 0; i < n; i++)
      if (gate[i] < 0) return fail(RS_ERR_INVALID, "%s[%d] = %d is negative", b->sched == RS_SCHED_NVS ? "required_rbs" : "data_to_transmit", i, gate[i]);
    memcpy(c->h_in + l.gate, gate, 4 * (size_t)n);
  }
  if (b->any_alpha) {
    bool need_hol = false;
    for (int i = 0; i < n; i++) {
      const int sl = h_slice[i];
      need_hol |= b->alpha[sl] && (b->sched == RS_SCHED_NVS || b->sched == RS_SCHED_NVS_NONGREEDY || b->beta[sl]);
    }
    if (need_hol && !in->hol_delay) return fail(RS_ERR_INVALID, "hol_delay is required by a customised (alpha=1, beta=1) slice");
    if (in->hol_delay) memcpy(c->h_in + l.hol, in->hol_delay, 8 * (size_t)n);
    else memset(c->h_in + l.hol, 0, 8 * (size_t)n);
    if (in->prio_has_data) memcpy(c->h_in + l.prio, in->prio_has_data, (size_t)n);
    else memset(c->h_in + l.prio, 1, (size_t)n);
  }
  hipStream_t st = b->stream;
  const clk::time_point t1 = c->timing ? clk::now() : clk::time_point();
  /* per-PRB reports and queue state are read again and again inside the TTI: those calls keep the device copies */
  const bool zc = c->z_in != nullptr && !in->cqi_prb && !b->any_alpha;
  uint8_t* const dev_in = zc ? c->z_in : c->d_in;
  uint8_t* const dev_out = zc ? c->z_out : c->d_out;
  if (!zc) {
    /* (an unchanged report set: the grid and the per-PRB block are on the device already; the per-call words lie between them) */
    if (reuse_grid) HIP_TRY(hipMemcpyAsync(c->d_in + l.slice, c->h_in + l.slice, l.prb - l.slice, hipMemcpyHostToDevice, st));
    else HIP_TRY(hipMemcpyAsync(c->d_in, c->h_in, in_bytes, hipMemcpyHostToDevice, st));
  }
  RsLaunch L = b->base;
  L.U = n;
  L.Upad = upad_of(n);
  L.n_ttis = 1;
  L.direct = 1;
  L.rand0 = in->rand0;
  L.rand1 = in->rand1;
  L.cqi_mode = RS_CQI_EPOCHS;
  L.refresh = 1;
  L.epochs = dev_in + l.grid;
  L.grid_stride = (int64_t)l.slice;
  L.n_epochs = 1;
  L.user_slice = dev_in + l.slice;
  L.avg = (double*)(dev_in + l.avg);
  L.prb_cqi = in->cqi_prb ? dev_in + l.prb : nullptr;
  L.grid_image = c->d_img;
  L.image_mode = image_mode;
  L.queue_mode = b->any_alpha ? 1 : 0;
  L.hol = (const double*)(dev_in + l.hol);
  L.prio = dev_in + l.prio;
  L.draws = dev_in + l.draws;
  L.gate = gate ? (const int32_t*)(dev_in + l.gate) : nullptr;
  L.exact_scan = exact_scan ? 1 : 0;
  L.gen_exp = b->gen_exp ? 1 : 0;
  L.gen_num = b->d_gen_num;
  if (b->sched == RS_SCHED_PF) { L.n_seg = (n + RS_PF_SEG - 1) / RS_PF_SEG; L.n_items = R * L.n_seg; }
  L.log_tbs = (int32_t*)(dev_out + l.tbs);
  L.log_uinfo = (int32_t*)(dev_out + l.uinfo);
  L.log_map = (int16_t*)(dev_out + l.map);
  L.log_quota = (int16_t*)(dev_out + l.quota);
  L.log_target = (int16_t*)(dev_out + l.target);
  const bool want_upper = b->sched == RS_SCHED_UPPERBOUND && (out->upper_rbg || out->upper_user);
  L.log_upper = want_upper ? (int32_t*)(dev_out + l.upper) : nullptr;
  /* direct mode: the kernel clears its per-user outputs itself and reads the single grid on every call */
  /* rs_ctx_specialize: this context's own build of the one-TTI kernel -- its lean form for the plain call (per-RBG CQI, no customised
   * slices, no gates, exponents in {0, 1}, every input inside the FP32 filter's range), the general one otherwise */
  RsJitKernel* kd = b->jit;
  int which = 0; /* 0: general build, 1: lean build */
  if (b->jit_lean && !L.prb_cqi && !L.queue_mode && !L.gate && !L.exact_scan && !L.gen_exp && !L.log_upper && !L.synthetic) { kd = b->jit_lean; which = 1; }
  /* A run-time build that does not carry the self-check mark serves its first calls beside the built-in kernel (rs_ctx_jit_status):
   * same inputs, same slice state; every output field and the slice state left behind must agree. */
  const bool checked_call = kd != nullptr && c->chk_left[which] > 0;
  const size_t sstate_bytes = 8 * (size_t)S, chk_half = round_up((int)sstate_bytes, 256) + round_up((int)sizeof(RsCellScalars), 256);
  if (checked_call) {
    uint8_t* const before = c->d_chk;
    uint8_t* const after = c->d_chk + chk_half;
    HIP_TRY(hipMemcpyAsync(before, b->d_sstate, sstate_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(before + round_up((int)sstate_bytes, 256), b->d_scal, sizeof(RsCellScalars), hipMemcpyDeviceToDevice, st));
    RsLaunch Lb = L; /* the built-in kernel, its outputs into a block of its own */
    Lb.log_tbs = (int32_t*)(c->d_out2 + l.tbs);
    Lb.log_uinfo = (int32_t*)(c->d_out2 + l.uinfo);
    Lb.log_map = (int16_t*)(c->d_out2 + l.map);
    Lb.log_quota = (int16_t*)(c->d_out2 + l.quota);
    Lb.log_target = (int16_t*)(c->d_out2 + l.target);
    Lb.log_upper = want_upper ? (int32_t*)(c->d_out2 + l.upper) : nullptr;
    Lb.done_flag = nullptr;
    HIP_TRY(rs_launch_cells(&Lb, b->threads, st));
    HIP_TRY(hipMemcpyAsync(c->h_out2.data(), c->d_out2, l.out_total, hipMemcpyDeviceToHost, st));
    /* keep what it left, put back what it found: the specialised kernel starts from the same state */
    HIP_TRY(hipMemcpyAsync(after, b->d_sstate, sstate_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_sstate, before, sstate_bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(b->d_scal, before + round_up((int)sstate_bytes, 256), sizeof(RsCellScalars), hipMemcpyDeviceToDevice, st));
    /* (a call with new reports: BOTH kernels transpose the caller's block and store the image -- the same bytes when the build is right;
     * a wrong image would show in the checked calls that read it) */
  }
  const bool poll = zc && c->poll && !checked_call;
  volatile uint32_t* const h_flag = (volatile uint32_t*)(c->h_out + c->flag_off);
  if (poll) {
    if (++c->seq == 0) c->seq = 1; /* (0 is the word's initial value) */
    L.done_flag = (uint32_t*)(c->z_out + c->flag_off);
    L.done_seq = c->seq;
  }
  if (kd) HIP_TRY(rs_jit_launch(kd, &L, st));
  else HIP_TRY(rs_launch_cells(&L, b->threads, st));
  if (!zc) HIP_TRY(hipMemcpyAsync(c->h_out, c->d_out, l.out_total, hipMemcpyDeviceToHost, st));
  const clk::time_point t2 = c->timing ? clk::now() : clk::time_point();
  bool seen = false;
  if (poll) {
    /* the kernel's last store publishes the sequence number after its outputs (release, system scope): spin on the pinned word */
    const clk::time_point give_up = clk::now() + std::chrono::microseconds(c->poll_us);
    for (unsigned spins = 0;; ++spins) {
      if (__atomic_load_n((const uint32_t*)h_flag, __ATOMIC_ACQUIRE) == c->seq) { seen = true; break; }
      __builtin_ia32_pause();
      if ((spins & 255u) == 255u && clk::now() > give_up) break;
    }
    if (seen) {
      c->n_polled++;
      /* the stream's own bookkeeping is settled without waiting: every 64th call asks it, so that nothing piles up in the runtime */
      if ((c->n_polled & 63) == 0) (void)hipStreamQuery(st);
    } else {
      c->n_fallback++;
    }
  }
  if (!seen) HIP_TRY(hipStreamSynchronize(st));
  if (checked_call) {
    /* field by field: the first difference is the message */
    std::vector<double> ss_jit(S), ss_ref(S);
    HIP_TRY(hipMemcpy(ss_jit.data(), b->d_sstate, sstate_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ss_ref.data(), c->d_chk + chk_half, sstate_bytes, hipMemcpyDeviceToHost));
    const uint8_t *ho = c->h_out, *hr = c->h_out2.data();
    char what[200] = "";
    auto differ32 = [&](const char* name, size_t off, int count) {
      const int32_t *a = (const int32_t*)(ho + off), *r = (const int32_t*)(hr + off);
      for (int i = 0; i < count && !what[0]; i++)
        if (a[i] != r[i]) snprintf(what, sizeof what, "%s[%d] = %d, the built-in kernel's %d", name, i, a[i], r[i]);
    };
    auto differ16 = [&](const char* name, size_t off, int count) {
      const int16_t *a = (const int16_t*)(ho + off), *r = (const int16_t*)(hr + off);
      for (int i = 0; i < count && !what[0]; i++)
        if (a[i] != r[i]) snprintf(what, sizeof what, "%s[%d] = %d, the built-in kernel's %d", name, i, a[i], r[i]);
    };
    differ16("rbg_to_user (call position)", l.map, R);
    differ16("quota_rbgs", l.quota, S);
    differ16("target_rbs", l.target, S);
    differ32("user_tbs_bits", l.tbs, n);
    differ32("user_nprb | final_cqi << 16 | mcs << 24", l.uinfo, n);
    if (want_upper) differ32("upper lists", l.upper, S * R);
    for (int i = 0; i < S && !what[0]; i++)
      if (to_bits(ss_jit[i]) != to_bits(ss_ref[i])) snprintf(what, sizeof what, "slice state[%d] = %a, the built-in kernel's %a", i, ss_jit[i], ss_ref[i]);
    if (what[0]) {
      /* the build is wrong: drop it (and its cache file), serve this call and all later ones from the built-in kernel */
      rs_jit_reject(kd);
      if (b->jit && b->jit != kd) rs_jit_reject(b->jit); /* (one wrong build of a shape: neither is trusted) */
      if (b->jit_lean && b->jit_lean != kd) rs_jit_reject(b->jit_lean);
      b->jit = nullptr;
      b->jit_lean = nullptr;
      c->jit_dropped = true;
      c->chk_left[0] = c->chk_left[1] = 0;
      memcpy(c->h_out, hr, l.out_total);
      HIP_TRY(hipMemcpy(b->d_sstate, c->d_chk + chk_half, sstate_bytes, hipMemcpyDeviceToDevice));
      snprintf(c->jit_msg, sizeof c->jit_msg, "self-check of the specialised %s build, checked call %d: %s; the build is dropped, the built-in kernel serves this "
               "context (lint the code object: tools/lint_exec_restore.py)", which ? "lean" : "general", c->chk_agreed[which] + 1, what);
      snprintf(g_err, sizeof g_err, "%s", c->jit_msg);
    } else {
      c->chk_agreed[which]++;
      if (--c->chk_left[which] == 0) rs_jit_mark_verified(kd);
    }
  }
  if (image_mode != 0) {
    /* the device holds this call's reports now */
    c->img_valid = true;
    c->img_epoch = in->cqi_epoch;
    c->img_n = n;
    c->img_prb = in->cqi_prb != nullptr;
    c->img_has_ids = in->user_id != nullptr;
    if (in->user_id && !reuse_grid) c->img_ids.assign(in->user_id, in->user_id + n);
    if (reuse_grid) c->n_img_reused++;
  }
  const clk::time_point t3 = c->timing ? clk::now() : clk::time_point();
  const int16_t* h_map = (const int16_t*)(c->h_out + l.map);
  const int16_t* h_quota = (const int16_t*)(c->h_out + l.quota);
  const int16_t* h_target = (const int16_t*)(c->h_out + l.target);
  const int32_t* h_tbs = (const int32_t*)(c->h_out + l.tbs);
  const int32_t* h_uinfo = (const int32_t*)(c->h_out + l.uinfo);
  for (int r = 0; r < R; r++) {
    int o = h_map[r];
    out->rbg_to_user[r] = o < 0 ? -1 : (in->user_id ? in->user_id[o] : o);
  }
  if (want_upper) {
    const int32_t* h_upper = (const int32_t*)(c->h_out + l.upper);
    for (int i = 0; i < S * R; i++) {
      const int32_t v = h_upper[i];
      if (out->upper_rbg) out->upper_rbg[i] = v < 0 ? -1 : (v & 63);
      if (out->upper_user) out->upper_user[i] = v < 0 ? -1 : (in->user_id ? in->user_id[v >> 8] : (v >> 8));
    }
  } else {
    if (out->upper_rbg) for (int i = 0; i < S * R; i++) out->upper_rbg[i] = -1;
    if (out->upper_user) for (int i = 0; i < S * R; i++) out->upper_user[i] = -1;
  }
  for (int s = 0; s < S; s++) {
    if (out->target_rbs) out->target_rbs[s] = h_target[s];
    if (out->quota_rbgs) out->quota_rbgs[s] = h_quota[s];
  }
  for (int i = 0; i < n; i++) {
    int32_t ui = h_uinfo[i];
    if (out->user_nprb) out->user_nprb[i] = ui & 0xFFFF;
    if (out->user_final_cqi) out->user_final_cqi[i] = (ui >> 16) & 0xFF;
    if (out->user_mcs) out->user_mcs[i] = (ui >> 24) & 0xFF;
    out->user_tbs_bits[i] = h_tbs[i];
  }
  if (c->timing) {
    const clk::time_point t4 = clk::now();
    auto us = [](clk::time_point a, clk::time_point b2) { return std::chrono::duration<double, std::micro>(b2 - a).count(); };
    c->t_prep += us(t0, t1);
    c->t_enq += us(t1, t2);
    c->t_wait += us(t2, t3);
    c->t_unpack += us(t3, t4);
    c->n_calls++;
  }
  return RS_OK;
}

/* Shape specialisation of a drop-in context (round 4): the one-TTI kernel compiled for this context's slices, RBGs, PRBs per RBG,
 * scheduler and user CAPACITY (the LDS carve), the users of a call staying a launch argument.  About two seconds per shape and
 * process (cached); results are identical.  On failure the context keeps the kernels built into the library. */
int rs_ctx_specialize(rs_ctx* c) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  rs_batch* b = c->b;
  if (b->jit) return RS_OK;
  if (c->jit_dropped) return fail(RS_ERR_STATE, "%s", c->jit_msg);
  HIP_TRY(hipSetDevice(b->cfg.cell.device));
  const bool gate_scratch = b->sched == RS_SCHED_PF || b->sched == RS_SCHED_NVS;
  b->jit_wanted = true;
  b->jit = rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, b->sched, gate_scratch ? 1 : 0, 0, b->jit_msg, sizeof b->jit_msg, 1);
  if (!b->jit) {
    snprintf(c->jit_msg, sizeof c->jit_msg, "%s", b->jit_msg[0] ? b->jit_msg : "hiprtc build failed");
    return fail(RS_ERR_HIP, "%s", c->jit_msg);
  }
  b->jit_msg[0] = 0;
  c->jit_msg[0] = 0;
  /* ... and its lean form (the per-call options of the plain call as constants); without it the general build serves every call */
  const char* const e_on = getenv("RS_JIT_LEAN");
  if (!e_on || atoi(e_on) != 0) {
    char msg[512] = "";
    b->jit_lean = rs_jit_get(b->cfg.cell.device, b->S, b->U, b->R, b->G, b->threads, b->sched, gate_scratch ? 1 : 0, 0, msg, sizeof msg, 1 | 4);
  }
  /* how many calls each build serves beside the built-in kernel before it is trusted (see rs_ctx_jit_status in the header): none for a
   * build that came from the cache with the mark of an earlier process's check */
  int calls = 8;
  if (const char* e = getenv("RS_DROPIN_SELFCHECK_CALLS")) calls = atoi(e) > 0 ? atoi(e) : 0;
  const char* pol = getenv("RS_JIT_SELFCHECK");
  const bool never = pol && pol[0] == '0', always = pol && pol[0] == '2';
  c->chk_agreed[0] = c->chk_agreed[1] = 0;
  c->chk_left[0] = (!never && (always || !rs_jit_is_verified(b->jit))) ? calls : 0;
  c->chk_left[1] = (b->jit_lean && !never && (always || !rs_jit_is_verified(b->jit_lean))) ? calls : 0;
  return RS_OK;
}

int rs_ctx_jit_status(rs_ctx* c, char* msg, size_t msglen) {
  if (!c) return fail(RS_ERR_INVALID, "null context");
  rs_batch* b = c->b;
  if (msg && msglen) {
    if (c->jit_dropped || !b->jit) {
      snprintf(msg, msglen, "%s", c->jit_msg);
    } else {
      /* per build: how it earned (or is still earning) its trust */
      auto state = [&](RsJitKernel* k, int which, char* out, size_t n) {
        if (!k) snprintf(out, n, "not built");
        else if (c->chk_left[which] > 0) snprintf(out, n, "%d checked call(s) agreed with the built-in kernel field by field, %d to go", c->chk_agreed[which], c->chk_left[which]);
        else if (c->chk_agreed[which] > 0) snprintf(out, n, "verified (%d checked calls agreed with the built-in kernel field by field)", c->chk_agreed[which]);
        else if (rs_jit_is_verified(k)) snprintf(out, n, "carries the self-check mark of an earlier check (cache file)");
        else snprintf(out, n, "unchecked (RS_JIT_SELFCHECK=0)");
      };
      char g[160], l[160];
      state(b->jit, 0, g, sizeof g);
      state(b->jit_lean, 1, l, sizeof l);
      snprintf(msg, msglen, "general build: %s; lean build: %s", g, l);
    }
  }
  if (c->jit_dropped) return -2;
  return b->jit ? 1 : (b->jit_wanted ? -1 : 0);
}

int rs_get_slice_offset(rs_ctx* c, double* offset) {
  if (!c || !offset) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipStreamSynchronize(c->b->stream));
  HIP_TRY(hipMemcpy(offset, c->b->d_sstate, 8 * c->b->S, hipMemcpyDeviceToHost));
  return RS_OK;
}

int rs_set_slice_offset(rs_ctx* c, const double* offset) {
  if (!c || !offset) return fail(RS_ERR_INVALID, "null argument");
  HIP_TRY(hipStreamSynchronize(c->b->stream));
  HIP_TRY(hipMemcpy(c->b->d_sstate, offset, 8 * c->b->S, hipMemcpyHostToDevice));
  return RS_OK;
}

/* ---- the reference's CQI trace files (include/radiosaber_hip.h) ---- */

int rs_trace_read_mapping(const char* path, int32_t* trace_of_entry, int32_t max_entries) {
  if (!path || (!trace_of_entry && max_entries > 0)) return fail(RS_ERR_INVALID, "null argument");
  std::ifstream ifs(path);
  if (!ifs) return fail(RS_ERR_INVALID, "cannot open %s", path);
  /* ref: enb-mac-entity.cc:50-53 -- `while (ifs >> uid >> tid) push_back(tid)`: the user id column is not used */
  long long uid, tid;
  int n = 0;
  while (ifs >> uid >> tid) {
    if (tid < 0 || tid > 0x7fffffff) return fail(RS_ERR_RANGE, "%s: trace id %lld out of range", path, tid);
    if (n < max_entries) trace_of_entry[n] = (int32_t)tid;
    n++;
  }
  if (n == 0) return fail(RS_ERR_INVALID, "%s: no \"<user> <trace>\" pairs", path);
  return n;
}

int rs_trace_read_ue_log(const char* path, int32_t n_rows, int32_t nb_rbs, int32_t rbg_size, uint8_t* out_rbg,
                         uint8_t* out_prb) {
  if (!path) return fail(RS_ERR_INVALID, "null argument");
  if (n_rows < 1 || nb_rbs < 1 || rbg_size < 1 || nb_rbs % rbg_size)
    return fail(RS_ERR_INVALID, "n_rows=%d nb_rbs=%d rbg_size=%d", n_rows, nb_rbs, rbg_size);
  std::ifstream ifs(path);
  if (!ifs) return fail(RS_ERR_INVALID, "cannot open %s", path);
  const int R = nb_rbs / rbg_size;
  int mixed = 0;
  int cqi = 0; /* ref: enb-mac-entity.cc:175 -- declared outside both loops: a failed extraction keeps the last value */
  std::vector<int> row(nb_rbs);
  for (int i = 0; i < n_rows; i++) {
    std::string line;
    std::getline(ifs, line);
    const char* q = line.c_str();
    bool failed = false; /* the istringstream's failbit: once set, every later `>>` leaves cqi alone */
    for (int j = 0; j < nb_rbs; j++) {
      if (!failed) {
        while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n' || *q == '\v' || *q == '\f') q++;
        if (!*q) {
          failed = true; /* end of line while skipping blanks: value untouched */
        } else {
          char* end = nullptr;
          errno = 0;
          long v = strtol(q, &end, 10);
          if (end == q) { cqi = 0; failed = true; } /* not a number: C++11 num_get stores 0 and fails */
          else {
            if (errno == ERANGE || v < 0 || v > 255)
              return fail(RS_ERR_RANGE, "%s: line %d value %d = %ld does not fit a CQI byte", path, i + 1, j + 1, v);
            cqi = (int)v;
            q = end;
          }
        }
      }
      row[j] = cqi;
    }
    for (int r = 0; r < R; r++) {
      for (int k = 1; k < rbg_size; k++)
        if (row[r * rbg_size + k] != row[r * rbg_size]) { mixed++; break; }
      if (out_rbg) out_rbg[(size_t)i * R + r] = (uint8_t)row[r * rbg_size];
    }
    if (out_prb)
      for (int j = 0; j < nb_rbs; j++) out_prb[(size_t)i * nb_rbs + j] = (uint8_t)row[j];
  }
  return mixed;
}

int rs_trace_load_dir(const char* dir, int32_t n_traces, int32_t n_rows, int32_t nb_rbs, int32_t rbg_size,
                      uint8_t* out_rbg) {
  if (!dir || !out_rbg) return fail(RS_ERR_INVALID, "null argument");
  if (n_traces < 1 || rbg_size < 1 || nb_rbs < 1 || nb_rbs % rbg_size) return fail(RS_ERR_INVALID, "bad trace shape");
  const size_t per = (size_t)n_rows * (nb_rbs / rbg_size);
  long long mixed = 0;
  for (int t = 0; t < n_traces; t++) {
    std::string f = std::string(dir) + "/ue" + std::to_string(t) + ".log"; /* ref: enb-mac-entity.cc:172 */
    int rc = rs_trace_read_ue_log(f.c_str(), n_rows, nb_rbs, rbg_size, out_rbg + (size_t)t * per, nullptr);
    if (rc < 0) return rc;
    mixed += rc;
  }
  return mixed > 0x7fffffff ? 0x7fffffff : (int)mixed;
}

int rs_lds_bytes_per_cell(int n_slices, int n_users, int n_rbgs, int sched, int threads) {
  if (n_slices < 1 || n_slices > RS_MAX_SLICES || n_users < 1 || n_users > RS_MAX_USERS || n_rbgs < 1 || n_rbgs > RS_MAX_RBGS ||
      threads < 64 || threads > 1024 || threads % 64)
    return fail(RS_ERR_INVALID, "bad shape");
  return rs_carve(n_slices, n_users, n_rbgs, sched, threads).lds_bytes;
}

/* ---- measurement helper ---- */
int rs_hbm_copy_probe(int device, uint64_t bytes, int iters, double* copy_gbs) {
  if (!copy_gbs || iters < 1 || bytes < (1u << 20) || (bytes & 15)) return fail(RS_ERR_INVALID, "bad probe arguments");
  HIP_TRY(hipSetDevice(device));
  void *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = RS_OK;
  float ms = 0;
#define PROBE_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rc = fail(RS_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); goto done; } } while (0)
  PROBE_TRY(hipMalloc(&src, bytes));
  PROBE_TRY(hipMalloc(&dst, bytes));
  PROBE_TRY(hipMemset(src, 1, bytes));
  PROBE_TRY(hipEventCreate(&e0));
  PROBE_TRY(hipEventCreate(&e1));
  PROBE_TRY(rs_launch_copy_probe(src, dst, bytes, nullptr)); /* warm-up */
  *copy_gbs = 0;
  for (int trial = 0; trial < 3; trial++) { /* best of three: the rate wanders by 10-15 % between trials */
    PROBE_TRY(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; i++) PROBE_TRY(rs_launch_copy_probe(src, dst, bytes, nullptr));
    PROBE_TRY(hipEventRecord(e1, nullptr));
    PROBE_TRY(hipEventSynchronize(e1));
    PROBE_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double g = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9; /* bytes read + bytes written */
    if (g > *copy_gbs) *copy_gbs = g;
  }
done:
#undef PROBE_TRY
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  return rc;
}

}  // extern "C"
