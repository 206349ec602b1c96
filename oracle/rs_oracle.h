/*
 * rs_oracle.h -- CPU ORACLE for the RadioSaber per-TTI downlink RBG allocation path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain C++ restatement of the reference's algorithm
 * (single thread, libm + libstdc++ std::sort, FP64, built with -ffp-contract=off).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (radiosaber_amd/, include/) never links, imports or calls anything under oracle/.
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - AMC tables ............ compared with the reference's own copy compiled into
 *                             oracle/_ref/libref_unittest_eesm.so (unittest/test_effective_sinr.cpp)
 *                             and, in the build container, with AMCModule.cpp's text.
 *   - EESM .................. compared with GetEesmEffectiveSinr compiled from
 *                             src/utility/eesm-effective-sinr.h (oracle/_ref/libref_eesm.so).
 *   - MaximizeCell/Vogel .... compared with the reference's static functions compiled from
 *                             unittest/test_tp_algos.cpp (oracle/_ref/libref_tp_algos.so).
 *   - whole sched-9 TTI loop  SURVEY.md Appendix A known-answer values (first scheduled TTI map,
 *                             cumu_bytes/cumu_rbs after 200 TTIs), tests/golden/appendix_a.json.
 *   - SubOpt ................ restated from the cited lines with the real std::unordered_map; PARITY UNPINNED
 *                             (no CLI number selects it, not in the unit program).
 *   - sched 1/7/8 loops ..... restated from the cited lines; PARITY UNPINNED beyond the shared
 *                             components above (the full simulator cannot be built in this image:
 *                             it needs jsoncpp and a generated header).
 *
 * All `ref:` citations are relative to /root/reference/src/.
 */
#ifndef RS_ORACLE_H_
#define RS_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* inter-slice policies, numbered like the reference CLI `sched` argument
 * (ref: scenarios/single-cell-with-interference.h:94-123) */
enum {
  RSO_SCHED_PF = 1,         /* DL_PF_PacketScheduler  -> DownlinkPacketScheduler::RBsAllocation */
  RSO_SCHED_NVS = 7,        /* DownlinkNVSScheduler                                            */
  RSO_SCHED_SEQUENTIAL = 8, /* DownlinkTransportScheduler, inter_sched_=0 GreedyByRow          */
  RSO_SCHED_MAXCELL = 9,    /* DownlinkTransportScheduler, inter_sched_=2 MaximizeCell         */
  RSO_SCHED_UPPERBOUND = 10,/* inter_sched_=4 UpperBound (not in the sweep)                     */
  RSO_SCHED_NVS_NONGREEDY = 11, /* DownlinkNVSScheduler with is_nongreedy_: RBsAllocationNonGreedyPF          */
  RSO_SCHED_SUBOPT = 101,   /* inter_sched_=1 SubOpt   (no CLI number reaches it)               */
  RSO_SCHED_VOGEL = 103     /* inter_sched_=3 VogelApproximate (no CLI number reaches it)       */
};

/* ---- AMC / EESM (ref: protocolStack/mac/AMCModule.cpp, utility/eesm-effective-sinr.h) ---- */
const int* rso_tbs_table(void);          /* [110][27] */
const int* rso_mcs_to_itbs(void);        /* [29] */
const int* rso_cqi_to_mcs(void);         /* [15] */
const double* rso_sinr_for_cqi(void);    /* [15] */
double rso_efficiency_from_cqi(int cqi);             /* AMCModule.cpp:320-327 */
int rso_cqi_from_sinr(double sinr_db);               /* AMCModule.cpp:253-261 */
int rso_tbs_bits(int mcs, int nb_rbs);               /* AMCModule.cpp:306-317 incl. the -O0 T[-1] rule */
double rso_eesm_effective_sinr(const double* sinr_db, int n); /* eesm-effective-sinr.h:33-46 */
int rso_rbg_size(int nb_rbs);                        /* eesm-effective-sinr.h:82-103; -1 if > 512 */
/* final CQI of an allocation: per-PRB CQIs in allocation order -> EESM -> CQI */
int rso_final_cqi(const uint8_t* cqi_per_prb, int n);

/* ---- glibc TYPE_3 rand() restatement (third-party: glibc 2.35 stdlib/random_r.c) ---- */
typedef struct { int32_t r[34]; int f, b; } rso_rng;
void rso_srand(rso_rng* g, unsigned seed);
int rso_rand(rso_rng* g);

/* ---- inter-slice assignment on an R x S efficiency grid (row-major [R][S]) ---- */
/* ref: downlink-transport-scheduler.cpp:249-272 / :351-376 / :378-451 / :274-349 */
void rso_greedy_by_row(const double* eff, const int* quota, int R, int S, int* rbg_to_slice);
void rso_maximize_cell(const double* eff, const int* quota, int R, int S, int* rbg_to_slice);
void rso_vogel(const double* eff, const int* quota, int R, int S, int* rbg_to_slice);
void rso_subopt(const double* eff, const int* quota, int R, int S, int* rbg_to_slice); /* :274-349 */
/* the post-std::sort order MaximizeCell scans (index = rbg*S+slice), for kernel unit tests */
void rso_maximize_cell_order(const double* eff, int R, int S, int* order);

/* ---- one cell ---- */
typedef struct rso_cell rso_cell;

typedef struct {
  int n_slices;              /* S */
  int n_users;               /* U = total UEs of the cell (user ids 0..U-1) */
  int n_rbgs;                /* R */
  int rbg_size;              /* PRBs per RBG; nb_rbs = R*rbg_size */
  int sched;                 /* RSO_SCHED_* */
  const double* weights;     /* [S] */
  const int* alpha;          /* [S] algo_alpha (1 needs rso_cell_set_queue_state; rso_cell_allocate only) */
  const int* beta;           /* [S] */
  const int* epsilon;        /* [S] */
  const int* psi;            /* [S] */
  const int* user_to_slice;  /* [U] */
} rso_config;

/* per-TTI outputs (all caller-allocated) */
typedef struct {
  int* target_rbs;     /* [S]  slice_target_rbs (sched 8/9/10), else 0 */
  int* quota_rbgs;     /* [S]  slice_quota_rbgs                        */
  int* rbg_to_user;    /* [R]  user id owning the RBG, -1 if none      */
  int* user_nprb;      /* [U]  PRBs allocated                          */
  int* user_final_cqi; /* [U]  0 if not scheduled                      */
  int* user_mcs;       /* [U]  */
  int* user_tbs_bits;  /* [U]  */
  int served_slice;    /* sched 7: slice chosen by SelectSliceToServe, else -1 */
  int* upper_rbg;      /* optional (may be NULL), sched 10: [S][R] RBGs every slice took, in push order, -1 padded */
  int* upper_user;     /* optional, sched 10: [S][R] the user each of them went to                            */
  double* slice_eff;   /* optional (may be NULL), sched 8/9/10/101/103: [R][S] flow_spectraleff, what the inter-slice step reads */
  int* slice_user;     /* optional with slice_eff: [R][S] user_index (-1: the slice has no user)                */
} rso_tti_out;

rso_cell* rso_cell_create(const rso_config* cfg);
void rso_cell_destroy(rso_cell* c);
/* current per-RBG CQI grid [U][R] (values 1..15); the oracle expands it to per-PRB internally */
void rso_cell_set_cqi(rso_cell* c, const uint8_t* cqi);
void rso_cell_set_user_cqi(rso_cell* c, int user, const uint8_t* cqi_row);
/* per-PRB CQI [U][R*rbg_size] (may vary inside an RBG, as the simulated channel's reports do) */
void rso_cell_set_cqi_prb(rso_cell* c, const uint8_t* cqi_prb);
/* bearer creation instant (RadioBearer ctor -> ResetTransmittedBytes: lastUpdate = Now) */
void rso_cell_set_last_update(rso_cell* c, double t);
/* the reference built with FIRST_SYNTHETIC_EXP / SECOND_SYNTHETIC_EXP (CONFIG/global_config:57-58, off as shipped): schedulers 7, 8,
 * 9, 10, 101, 103 size the transport block PRB by PRB, each with the MCS of its own CQI
 * (downlink-transport-scheduler.cpp:653-659, downlink-nvs-scheduler.cpp:336-342) */
void rso_cell_set_synthetic_exp(rso_cell* c, int on);
/* alpha != 0 slices: head-of-line delay and "prioritized bearer has data" per user */
void rso_cell_set_queue_state(rso_cell* c, const double* hol, const uint8_t* prio_has_data);
/* one TTI of DoSchedule(): EWMA update at time `now`, RBsAllocation with the two rand() values,
 * DoStopSchedule accounting.  active==NULL: every user is backlogged. */
int rso_cell_step(rso_cell* c, double now, int rand0, int rand1, rso_tti_out* out);
/* RBsAllocation() alone on caller-provided PF state (no EWMA / accounting): mirrors rs_schedule_tti */
int rso_cell_allocate(rso_cell* c, const double* avg_rate, int rand0, int rand1, rso_tti_out* out);
/* sched 11: RBsAllocationNonGreedyPF for the users of `slice` with the rand() values it would draw, in draw order
 * (RSO_NONGREEDY_SAMPLES x users-of-the-slice values); and the whole DoSchedule() drawing from a generator */
#define RSO_NONGREEDY_SAMPLES 300
int rso_cell_allocate_nongreedy(rso_cell* c, const double* avg_rate, int slice, const int* draws, int n_draws,
                                rso_tti_out* out);
int rso_cell_step_rng(rso_cell* c, double now, rso_rng* g, rso_tti_out* out);
/* state access */
void rso_cell_get_state(const rso_cell* c, double* avg_rate, int64_t* cum_bytes, int64_t* cum_rbs,
                        double* slice_offset_or_ewma);
void rso_cell_set_avg_rate(rso_cell* c, const double* avg_rate);
void rso_cell_set_slice_offset(rso_cell* c, const double* offset); /* slice_rbs_offset_ [S] */
/* users holding two bearers in this TTI (MAX_BEARERS = 2, packet-scheduler.h:31): avg_rate is the lower-priority
 * bearer's average, avg2[u] the other one's (< 0: one bearer); NULL clears.  Summed as the reference does, (1 + a) + a2. */
void rso_cell_set_second_bearer_avg(rso_cell* c, const double* avg2);

/* ---- stand-alone runs (the simulator cadence restated; ref: core/eventScheduler/simulator.cc:117-126,
 *      componentManagers/FrameManager.cpp:158-189, device/CqiManager/cqi-manager.cpp:94-123,
 *      protocolStack/mac/enb-mac-entity.cc:160-193, phy/wideband-cqi-eesm-error-model.cpp:69) ---- */
typedef struct {
  const uint8_t* trace;   /* [n_traces][n_rows][R] per-RBG CQI */
  int n_traces, n_rows;
  const int* mapping;     /* [n_map] trace id of (user_id % n_map) */
  int n_map;
  int row_modulus;        /* 475 in the reference (MAX_TTI_TRACE); rows >= n_rows must not be hit */
  unsigned seed;          /* srand() argument (seed.h commonSeed[i]) */
  long rand_skip;         /* rand() values consumed before the scheduler's first draw */
  int phy_error_draws;    /* 1: one rand() per UE served in the previous TTI (error model) */
  int first_tti;          /* 100: applications start at 0.1 s */
  int n_ttis;             /* scheduled TTIs to run */
} rso_trace_run;

/* Runs n_ttis scheduled TTIs.  If log_* are non-NULL they receive per-TTI rows:
 * log_rbg_to_user [n_ttis][R], log_final_cqi [n_ttis][U], log_quota [n_ttis][S], log_target [n_ttis][S],
 * log_tbs_bits [n_ttis][U].  Final counters via rso_cell_get_state. */
int rso_run_trace(rso_cell* c, const rso_trace_run* run, int* log_rbg_to_user, int* log_final_cqi,
                  int* log_quota, int* log_target, int* log_tbs_bits);

/* synthetic-grid run used by the benchmark and the batched parity tests:
 * CQI grid redrawn every 40 TTIs from cqi_epochs [n_epochs][U][R]; per-cell rand stream seeded
 * with `seed`, two draws per TTI (+ the error-model draws if phy_error_draws). */
int rso_run_synth(rso_cell* c, const uint8_t* cqi_epochs, int n_epochs, int refresh, unsigned seed,
                  int phy_error_draws, int n_ttis, int* log_rbg_to_user, int* log_tbs_bits);

/* the two runs with per-PRB sources: cqi_prb_epochs [n_epochs][U][R*rbg_size]; run->trace [n_traces][n_rows][R*rbg_size] */
int rso_run_synth_prb(rso_cell* c, const uint8_t* cqi_prb_epochs, int n_epochs, int refresh, unsigned seed,
                      int phy_error_draws, int n_ttis, int* log_rbg_to_user, int* log_tbs_bits);
int rso_run_trace_prb(rso_cell* c, const rso_trace_run* run, int* log_rbg_to_user, int* log_final_cqi,
                      int* log_quota, int* log_target, int* log_tbs_bits);

/* ---- finite queues.  PARITY UNPINNED (restated from flows/MacQueue.cpp, protocolStack/rlc/um-rlc-entity.cpp,
 *      flows/radio-bearer.cpp:281-367, downlink-transport-scheduler.cpp:105-221, packet-scheduler.cpp:305-335). ----
 * bearer_kind [U][2], index = bearer priority: 0 none, 1 InfiniteBuffer, 2 finite queue fed by arrival bursts */
#define RSO_FULL_PACKET 1495 /* MAXMTUSIZE 1490 + UDP 8 + IP 20, ROHC 28 -> 3, PDCP 2 (protocolStack/packet/Packet.cpp:84-118) */
void rso_cell_enable_queues(rso_cell* c, const uint8_t* bearer_kind);
/* arrival bursts of one bearer, ascending in time: at time[i] the application enqueues n_full[i] packets of RSO_FULL_PACKET
 * bytes and then, if last[i] > 0, one packet of last[i] bytes */
void rso_cell_set_arrivals(rso_cell* c, int user, int prio, int n, const double* time, const int32_t* n_full, const int32_t* last);
/* one DoSchedule() with queues: arrivals up to `now`, EWMA of every bearer, the users' records, RBsAllocation (transport
 * schedulers and sched 7), DoStopSchedule incl. the RLC dequeue */
int rso_cell_step_queues(rso_cell* c, double now, rso_rng* g, rso_tti_out* out); /* draws 2 values when it allocates */
int rso_run_synth_queues(rso_cell* c, const uint8_t* cqi_epochs, int n_epochs, int refresh, unsigned seed, int n_ttis,
                         int* log_rbg_to_user, int* log_tbs_bits);
/* the same run on per-PRB grids ([U][R*G] per epoch) */
int rso_run_synth_queues_prb(rso_cell* c, const uint8_t* cqi_prb_epochs, int n_epochs, int refresh, unsigned seed, int n_ttis,
                         int* log_rbg_to_user, int* log_tbs_bits);
/* per bearer [U][2]: PF average, cumulative bytes / RBs, MAC queue bytes and packets; any pointer may be NULL */
void rso_cell_get_bearer_state(const rso_cell* c, double* avg, int64_t* cum_bytes, int64_t* cum_rbs, int32_t* queue_bytes,
                               int32_t* queue_packets);

/* the simulated clock at the start of scheduled TTI first_tti + k, k = 0..n-1 (simulator.cc:117-126) */
void rso_clock_ticks(int first_tti, int n, double* out);

/* many independent cells over the host cores (OpenMP): the cpu_baseline of bench.py */
int rso_run_synth_many(const rso_config* cfg, int n_cells, const uint8_t* cqi_epochs, int n_epochs, int refresh,
                       const unsigned* seeds, int phy_error_draws, int n_ttis, int threads, int64_t* total_bytes,
                       int* threads_used);
/* the same with per-cell epochs [n_cells][n_epochs][U][R] and every cell's final state returned (bench.py's parity sample) */
int rso_run_synth_cells(const rso_config* cfg, int n_cells, const uint8_t* cqi_epochs, int n_epochs, int refresh, const unsigned* seeds,
                        int phy_error_draws, int n_ttis, int threads, double* avg_rate, int64_t* cum_bytes, int64_t* cum_rbs,
                        double* slice_state, int* threads_used);

#ifdef __cplusplus
}
#endif
#endif /* RS_ORACLE_H_ */
