/*
 * radiosaber_hip.h -- C ABI of the MI355X-native RadioSaber downlink RBG allocation path.
 *
 * Plain C, caller-owned flat buffers, int status returns, no exceptions, no torch types.
 * One context per host thread / HIP stream.  Implemented by libradiosaber_hip.so
 * (radiosaber_amd/csrc/, hand-written HIP for gfx950; there is NO CPU fallback: every entry point
 * that computes returns RS_ERR_NO_DEVICE / RS_ERR_HIP when no MI355X is usable).
 *
 * The reference (elvinlife/RadioSaber, an LTE-Sim fork) has no C ABI or plug-in loader: its
 * boundary is the C++ virtual seam
 *     PacketScheduler::Schedule() -> virtual DoSchedule() -> virtual RBsAllocation()
 *     (src/protocolStack/mac/packet-scheduler/packet-scheduler.cpp:72-90, packet-scheduler.h:133-137)
 * installed by ENodeB::SetDLScheduler (src/device/ENodeB.cpp:303-391).  Each entry point below
 * names the reference function(s) it replaces; INTEGRATION.md shows the C++ adapter class a
 * maintainer adds to the LTE-Sim tree to call them.  `ref:` paths are relative to the reference's
 * src/protocolStack/mac/packet-scheduler/ unless they start with src/.
 */
#ifndef RADIOSABER_HIP_H_
#define RADIOSABER_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ABI_VERSION 11 /* 11: rs_config.link_tables (RS_LINK_*) + rs_link_tables_pinned / rs_link_tables_compare, rs_tti_in.cqi_epoch (the context keeps the
                               CQI image of an unchanged report set on the device), rs_ctx_jit_status (a specialised context checks its run-time build against the
                               built-in kernel during its first calls), rs_batch_config.selfcheck -1 / 0 / 1 with run-time builds verified by default and the
                               self-check mark in the cache file, rs_jit_compiler_identity (the compiler's full identity in the cache key);
                            10: rs_batch_debug_heap_sorts / rs_ctx_debug_heap_sorts (the sort emulation's heap-sort fallback counted per device site),
                               rs_jit_cache_stats / _file / _warm (code objects cached on disk), rs_batch_config.autotune + rs_batch_autotune_report, rs_batch_config.selfcheck, rs_batch_debug_clocks, rs_batch_checkpoint_bytes / _save / _load;
                            9: rs_create_checked / rs_batch_create_checked (RS_CREATE / RS_BATCH_CREATE: the caller's ABI version and struct size are
                            checked), rs_batch_config.cqi_epoch_wrap / queue_state_lds, threads_per_cell up to 1024 with jit, rs_jit_selfcheck_untuned, rs_batch_write_state, rs_ctx_specialize, rs_jit_selfcheck_dropin;
                            8: rs_jit_selfcheck_queue, rs_config.synthetic_exp, any integer algo_epsilon / algo_psi in drop-in contexts; 7: rs_device_source_hash; rs_schedule_tti accepts any double as avg_rate / hol_delay (exact scan outside the FP32 filter's range);
                            6: rs_tti_in.required_rbs / data_to_transmit (the gates of schedulers 7 and 1 in the drop-in mode);
                            5: per-PRB batch sources (rs_batch_upload_cqi_epochs_prb, rs_batch_set_trace_prb);
                            4: finite queues in batches (rs_batch_set_bearers, rs_batch_set_arrivals, rs_batch_read_bearer_state, rs_internet_flow_arrivals);
                            2: rs_tti_in.rand_draws, schedulers 10 / 11 / 101 / 103, rs_trace_*, rs_hbm_copy_probe, rs_lds_bytes_per_cell;
                            3: rs_get_rbg_size, rs_dl_prbs_for_bandwidth, rs_batch_read_clock, rs_batch_jit_status,
                               rs_batch_synthesize_cqi_at, rs_batch_run_logged_ex */

/* status codes */
enum {
  RS_OK = 0,
  RS_ERR_INVALID = -1,    /* bad argument / unsupported configuration (message: rs_last_error) */
  RS_ERR_NO_DEVICE = -2,  /* no usable HIP device */
  RS_ERR_HIP = -3,        /* a HIP runtime call failed */
  RS_ERR_STATE = -4,      /* call order (e.g. run before a CQI source is set) */
  RS_ERR_RANGE = -5       /* device-side check failed (trace row out of range, ...) */
};

/* scheduler selection, numbered like the reference CLI `sched` argument
 * (ref: src/scenarios/single-cell-with-interference.h:94-123, src/device/ENodeB.h:66-81) */
enum {
  RS_SCHED_PF = 1,         /* DL_PF_PacketScheduler -> DownlinkPacketScheduler::RBsAllocation
                              (ref: downlink-packet-scheduler.cpp:179-331, dl-pf-packet-scheduler.cpp:128-140) */
  RS_SCHED_NVS = 7,        /* DownlinkNVSScheduler (ref: downlink-nvs-scheduler.cpp:94-142,275-358) */
  RS_SCHED_SEQUENTIAL = 8, /* DownlinkTransportScheduler + GreedyByRow (ref: downlink-transport-scheduler.cpp:249-272) */
  RS_SCHED_MAXCELL = 9,    /* DownlinkTransportScheduler + MaximizeCell = RadioSaber (ref: :351-376) */
  RS_SCHED_NVS_NONGREEDY = 11, /* DownlinkNVSScheduler with is_nongreedy_ (the CLI's scheduler 11): RBsAllocationNonGreedyPF +
                              AssignRBsGivenMCS (ref: downlink-nvs-scheduler.cpp:405-528), 300 sampled CQI-index vectors per TTI */
  RS_SCHED_UPPERBOUND = 10, /* DownlinkTransportScheduler + UpperBound (ref: :223-246, apply step :603-616): every slice takes
                              its own best quota RBGs whatever the others take -- an upper bound, not an allocation: several
                              UEs may hold one RBG.  rbg_to_user then reports the UE of the lowest-numbered slice holding
                              the RBG; user_nprb / user_tbs_bits / ... are complete and rs_tti_out.upper_rbg / upper_user list
                              every slice's RBGs in push order.  Needs n_rbgs*n_slices <= 2048. */
  RS_SCHED_SUBOPT = 101,   /* DownlinkTransportScheduler + SubOpt (ref: :274-349; inter_sched_ = 1, which no CLI number selects):
                            * best slice per RBG, then least-loss moves from slices above to slices below their quota, ties
                            * in the order libstdc++'s unordered_map yields the slices */
  RS_SCHED_VOGEL = 103     /* DownlinkTransportScheduler + VogelApproximate (ref: :378-451; inter_sched_ = 3, which no CLI
                              scheduler number of the reference selects -- ENodeB::DLScheduler_VOGEL exists, ENodeB.cpp:375) */
};

#define RS_MAX_SLICES 64
#define RS_MAX_RBGS 64
#define RS_MAX_USERS 1024

/* What the reference scheduler constructors parse out of the JSON config
 * (ref: downlink-transport-scheduler.cpp:55-97, downlink-nvs-scheduler.cpp:44-86,
 *  dl-pf-packet-scheduler.cpp:40-58) plus the cell's PRB grid
 * (ref: RBsAllocation :457-461 nb_rbs/rbg_size; src/utility/eesm-effective-sinr.h:82-103). */
typedef struct rs_config {
  int32_t n_slices;             /* S = ues_per_slice.size()                      (1..64)   */
  int32_t n_users;              /* U = sum(ues_per_slice); user ids are 0..U-1   (1..1024) */
  int32_t n_rbgs;               /* R = nb_rbs / rbg_size                         (1..64)   */
  int32_t rbg_size;             /* PRBs per RBG = get_rbg_size(nb_rbs)           (1..8)    */
  int32_t sched;                /* RS_SCHED_*                                               */
  int32_t device;               /* HIP device ordinal                                       */
  const double* slice_weight;   /* [S] "weight"                                             */
  const int32_t* algo_alpha;    /* [S] 0, or 1 = customised slice (ref: :694-711): the queue state comes in through
                                   rs_tti_in.hol_delay / prio_has_data (drop-in mode) or from the batch's own queue model
                                   (rs_batch_set_bearers / rs_batch_set_arrivals)                       */
  const int32_t* algo_beta;     /* [S] 0 or 1: with alpha = 1, multiply the metric by the HoL delay  */
  const int32_t* algo_epsilon;  /* [S] exponent of the rate in pow(se_kbps, epsilon) / pow(avg_kbps, psi) (ref: downlink-transport-
                                 * scheduler.cpp:690-693).  rs_create (drop-in): any integer in -64..64 -- the host's libm raises
                                 * the powers (16 numerators per slice once, every user's denominator per call), the device divides
                                 * and compares exactly.  rs_batch_create: 0 or 1 only (pow(x,0)=1, pow(x,1)=x are exact; the PF
                                 * averages live on the device, which has no pow())                                        */
  const int32_t* algo_psi;      /* [S] exponent of the average, same rule                                                  */
  const int32_t* user_to_slice; /* [U] non-decreasing (run-length expansion of ues_per_slice) */
  void* stream;                 /* hipStream_t to launch on, NULL = a stream owned by the context */
  int32_t synthetic_exp;        /* 0 (as shipped), or 1 = the reference built with FIRST_SYNTHETIC_EXP / SECOND_SYNTHETIC_EXP
                                 * (CONFIG/global_config:57-58): schedulers 7, 8, 9, 10, 101, 103 size a transport block PRB by
                                 * PRB, each with the MCS of its own CQI (downlink-transport-scheduler.cpp:653-659,
                                 * downlink-nvs-scheduler.cpp:336-342; the PDCCH record keeps the EESM MCS); schedulers 1 and 11 have
                                 * no such branch and ignore it.  (ABI 8)                                                      */
  int32_t link_tables;          /* RS_LINK_*: where the EESM constants E[c] / X[k] come from (see rs_link_tables).  (ABI 11)    */
} rs_config;

/* The EESM constants are transcendental: the reference evaluates exp / pow / log / log10 with the libm of the machine it was built on
 * (src/utility/eesm-effective-sinr.h:33-46, src/protocolStack/mac/AMCModule.cpp:253-261), and a final CQI can differ by one between two
 * libms that round one of them differently.
 *   RS_LINK_PINNED_GLIBC_2_35  the values glibc 2.35 (x86-64) gives -- the libm behind every fixture of this repository (SURVEY.md
 *                              Appendix A, tests/golden/appendix_a.json) -- compiled into the library as data: results do not depend
 *                              on the machine that runs the GPU
 *   RS_LINK_HOST_LIBM          this host's libm, evaluated at create time exactly as the reference evaluates them: what a drop-in needs
 *                              beside a reference built on the same machine
 *   RS_LINK_DEFAULT (0)        batches: pinned; drop-in contexts (rs_create): host libm -- and when the two sets differ on this host,
 *                              rs_create still succeeds and rs_last_error() carries one line that says where (empty otherwise) */
enum { RS_LINK_DEFAULT = 0, RS_LINK_HOST_LIBM = 1, RS_LINK_PINNED_GLIBC_2_35 = 2 };

const char* rs_last_error(void);   /* thread-local message of the last failing call */
/* The structs of this header grow at their ends from one ABI version to the next, and the plain create functions below read them
 * with THIS library's layout.  A caller that may meet a library of another version either compares rs_abi_version() with the
 * RS_ABI_VERSION it was compiled with before creating anything, or creates through RS_CREATE / RS_BATCH_CREATE
 * (rs_create_checked / rs_batch_create_checked), which pass the caller's version and struct size and are refused with
 * RS_ERR_INVALID on any mismatch -- nothing is inferred from field values. */
int rs_abi_version(void);
int rs_device_count(void);         /* number of HIP devices (0 when none)           */

/* Link-adaptation constants the device uses, computed by the HOST libm exactly as the reference
 * evaluates them (ref: src/utility/eesm-effective-sinr.h:33-46, src/protocolStack/mac/AMCModule.cpp:253-261,320-327):
 *   eff[c]  = (TBS(1 PRB, mcs(c)) / 0.001) / 180000.            c = 1..15  (eff[0] = 0)
 *   kbps[c] = eff[c] * 180000 / 1000
 *   E[c]    = exp(-pow(10, SINRForCQIIndex[c-1] / 10))
 *   X[k]    = max{ x : 10*log10(-1*log(x)) >= SINRForCQIIndex[k] }   k = 1..13
 * final CQI of an allocation = 1 + #{k : x <= X[k]}, x = (sum of E over its PRBs) / nPRB; x == 0 -> 15.
 * Needs no GPU.  Returns RS_ERR_INVALID if the host libm is not monotone around a threshold. */
int rs_link_tables(double eff[16], double kbps[16], double eesm_e[16], double eesm_x[16]);
/* the same with E[c] / X[k] from the pinned glibc-2.35 set (eff / kbps are IEEE divisions of table integers: the same on any host) */
int rs_link_tables_pinned(double eff[16], double kbps[16], double eesm_e[16], double eesm_x[16]);
/* this host's libm against the pinned set: the number of E / X entries that differ (0: they agree), msg receives "E[c] host ... pinned
 * ..." for each; RS_ERR_INVALID when the host's libm is not monotone around a threshold.  Needs no GPU. */
int rs_link_tables_compare(char* msg, size_t msglen);

/* replaces get_rbg_size() (ref: src/utility/eesm-effective-sinr.h:82-103): PRBs per RBG of a cell with nb_rbs PRBs
 * (<= 10: 1, <= 26: 2, <= 63: 3, <= 110: 4, <= 512: 8).  Above 512 the reference throws std::runtime_error: RS_ERR_INVALID. */
int rs_get_rbg_size(int nb_rbs);
/* replaces BandwidthManager's GetDlSubChannels().size() (ref: src/core/spectrum/bandwidth-manager.cpp:30-38, 52-108):
 * 1.4 -> 6, 3 -> 15, 5 -> 25, 10 -> 50, 15 -> 75, 20 -> 100, 100 -> 512 PRBs, anything else 25 as the reference's else-branch. */
int rs_dl_prbs_for_bandwidth(double bw_mhz);

/* ------------------------------------------------------------------------------------------
 * Drop-in mode: one RBsAllocation() call.
 * ------------------------------------------------------------------------------------------ */
typedef struct rs_ctx rs_ctx;

/* replaces the scheduler constructor + ENodeB::SetDLScheduler (ref: src/device/ENodeB.cpp:303-391) */
rs_ctx* rs_create(const rs_config* cfg);
rs_ctx* rs_create_checked(const rs_config* cfg, int abi_version, size_t cfg_size);
#define RS_CREATE(cfg) rs_create_checked((cfg), RS_ABI_VERSION, sizeof(rs_config))
void rs_destroy(rs_ctx* ctx);

/* What RBsAllocation() sees on entry (ref: GetUsersToSchedule(): packet-scheduler.h:88-123):
 * the users with queued data, in first-seen (= ascending user id) order. */
typedef struct rs_tti_in {
  int32_t n_users;          /* users to schedule this TTI (<= cfg.n_users)                          */
  const int32_t* user_id;   /* [n] ascending user ids; NULL = 0..n-1                                */
  const uint8_t* cqi;       /* [n][R] CQI (1..15) of PRB rbg*rbg_size = GetCqiFeedbacks().at(rbg*rbg_size);
                               the whole RBG carries that CQI (true for every shipped trace);
                               see cqi_prb for the general case                                     */
  const double* avg_rate;   /* [n] GetAverageTransmissionRate() of the user's bearer; the kernel forms the reference's
                             *     `averageRate = 1 + avg` (:681-686).  A user holding two bearers (MAX_BEARERS = 2): pass
                             *     ((1 + a0) + a1) - 1, which is exact for averages >= 1 and reproduces the reference's
                             *     summation order bit for bit (the C++ adapter does this) */
  int32_t rand0, rand1;     /* the two rand() values RBsAllocation draws (ref: :490, :511);
                               ignored by RS_SCHED_PF / RS_SCHED_NVS                                */
  const uint8_t* cqi_prb;   /* optional [n][R*rbg_size]: the full per-PRB GetCqiFeedbacks() vectors.  When
                               non-NULL `cqi` is ignored (the metric reads PRB rbg*rbg_size, ref: :536) and link
                               adaptation reads every allocated PRB (ref: :643-646), so CQIs may differ inside
                               an RBG as the simulated channel's reports do                           */
  /* customised slices (algo_alpha = 1), ref: downlink-transport-scheduler.cpp:694-711,
   * downlink-nvs-scheduler.cpp:375-387; both may be NULL when every slice has algo_alpha = 0 */
  const double* hol_delay;       /* [n] GetHeadOfLinePacketDelay() of the user's slice-priority bearer     */
  const uint8_t* prio_has_data;  /* [n] m_dataToTransmit[slice_priority_[slice]] != 0; NULL = all 1        */
  const int32_t* rand_draws;     /* RS_SCHED_NVS_NONGREEDY only: the 300 * n values rand() returns to
                                    RBsAllocationNonGreedyPF, in draw order (sample-major, user-minor;
                                    downlink-nvs-scheduler.cpp:431-441); NULL for every other scheduler     */
  /* finite queues: the two gates that never bind for InfiniteBuffer flows.  NULL = backlogged (no gate). */
  const int32_t* required_rbs;     /* RS_SCHED_NVS, [n]: UserToSchedule::m_requiredRBs in PRBs; a user competes for an RBG
                                      only while its allocated PRBs are below it (downlink-nvs-scheduler.cpp:299-300)       */
  const int32_t* data_to_transmit; /* RS_SCHED_PF, [n] bytes: FlowToSchedule::GetDataToTransmit(); a flow leaves the TTI's
                                      competition once the transport block of its PRBs so far carries data * 8 bits
                                      (downlink-packet-scheduler.cpp:253-265)                                               */
  uint64_t cqi_epoch;       /* 0: no promise -- the CQI block is validated, copied and read on every call.  Non-zero: the caller's
                               version number of `cqi` (or `cqi_prb`): it changes the number whenever ANY report changed.  The reference
                               refreshes a UE's CQI every 40 TTIs (src/protocolStack/mac/enb-mac-entity.cc:38 CQI_INTERVAL,
                               src/device/CqiManager/cqi-manager.cpp:115), so 39 calls of 40 see the grid of the call before.  A call
                               whose cqi_epoch, n_users and user_id list equal the previous call's does not touch the caller's block:
                               the kernel reads the image the context kept on the device (the grid in the layout of its LDS, HBM-resident)
                               instead of n * R bytes over the host link, and the host skips the range check and the copy.  (ABI 11) */
} rs_tti_in;

/* What RBsAllocation() leaves behind (ref: :589-620 allocation lists + slice_rbs_offset_,
 * :630-674 UpdateAllocatedBits / PDCCH records). All arrays caller-allocated. */
typedef struct rs_tti_out {
  int32_t* target_rbs;      /* [S] slice_target_rbs   (0 for PF/NVS)                      */
  int32_t* quota_rbgs;      /* [S] slice_quota_rbgs   (0 for PF/NVS)                      */
  int32_t* rbg_to_user;     /* [R] user id owning RBG r (its PRBs r*rbg_size..+rbg_size-1), -1 = none */
  int32_t* user_nprb;       /* [n] GetListOfAllocatedRBs()->size()                        */
  int32_t* user_final_cqi;  /* [n] GetCQIFromSinr(GetEesmEffectiveSinr(..)), 0 = not scheduled */
  int32_t* user_mcs;        /* [n] mcs of the PDCCH records                               */
  int32_t* user_tbs_bits;   /* [n] UpdateAllocatedBits() argument                         */
  /* RS_SCHED_UPPERBOUND only, both optional (NULL = not wanted): what the apply step :603-616 walks -- per slice the RBGs it
   * took in push order and the user id each one went to (user_index[rbg][slice]); [S][n_rbgs], -1 padded */
  int32_t* upper_rbg;
  int32_t* upper_user;
} rs_tti_out;

/* replaces DownlinkTransportScheduler::RBsAllocation (ref: downlink-transport-scheduler.cpp:453-675),
 * DownlinkPacketScheduler::RBsAllocation (ref: downlink-packet-scheduler.cpp:179-331) and
 * DownlinkNVSScheduler::RBsAllocation (ref: downlink-nvs-scheduler.cpp:275-358; pass the users of the
 * slice SelectSliceToServe chose).  The context carries slice_rbs_offset_ between calls -- and nothing else of the simulator's state:
 * PF averages, pending grants, cumulative counters, the clock and the rand() stream stay the caller's (DoSchedule / DoStopSchedule keep
 * them in the reference too); with rs_tti_in.cqi_epoch it also keeps an image of the last CQI grid on the device.
 * One context per host thread.  Many contexts in one process: export GPU_MAX_HW_QUEUES >= their number (up to 16) before the process
 * starts -- the HIP runtime maps a process's streams onto 4 hardware queues by default, and one-TTI kernels of contexts that share a
 * queue run one after the other; rs_create says so through rs_last_error() when it happens (profiles/r06_dropin_concurrency.md). */
int rs_schedule_tti(rs_ctx* ctx, const rs_tti_in* in, rs_tti_out* out);
/* Optional, once after rs_create: compile this context's own build of the one-TTI kernel (hiprtc, ~2 s per shape and process,
 * cached) -- slices, RBGs, PRBs per RBG, scheduler and the user capacity as compile-time constants, the users of a call still a
 * launch argument.  Results are identical; a call gets shorter.  Two builds (~2 s each): the general one and a lean
 * one that serves the plain call -- per-RBG CQI, no customised slices, no gates, exponents in {0, 1}, every input an ordinary
 * FP32 number -- with those per-call options as constants (RS_JIT_LEAN=0: the general one only).  RS_OK, or RS_ERR_HIP with the
 * context left on the kernels built into the library.  The C++ adapter calls it from its constructor.  (ABI 9) */
int rs_ctx_specialize(rs_ctx* ctx);
/* A specialised context does not trust its run-time build blindly (round 6): unless the code object came from the disk cache with the
 * self-check mark of an earlier process, the first RS_DROPIN_SELFCHECK_CALLS (default 8) calls each of its two builds serves run on the
 * kernel built into the library AS WELL, from the same slice state, and every field of rs_tti_out plus the slice state left behind is
 * compared.  Agreement for all of them leaves the mark in the cache file; one difference drops the build for good -- that call and all
 * later ones are served by the built-in kernel, the call still returns RS_OK with the built-in kernel's (correct) results, and
 * rs_last_error() / rs_ctx_jit_status say which field of which user differed.  RS_JIT_SELFCHECK=0 switches the check off, =2 checks
 * marked builds too.
 * rs_ctx_jit_status: 1 = the specialised kernels serve the calls (msg: empty, or how many checked calls agreed), 0 = rs_ctx_specialize
 * was not called, -1 = it failed to build, -2 = a build was dropped by the check (msg says what differed). */
int rs_ctx_jit_status(rs_ctx* ctx, char* msg, size_t msglen);
/* build check without a GPU: does that kernel compile for a context of this shape? (code size or a negative value) */
int rs_jit_selfcheck_dropin(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, char* err, size_t errlen);
/* slice_rbs_offset_ accessors (ref: downlink-transport-scheduler.h:38) */
int rs_get_slice_offset(rs_ctx* ctx, double* offset /* [S] */);
int rs_set_slice_offset(rs_ctx* ctx, const double* offset /* [S] */);

/* ------------------------------------------------------------------------------------------
 * Batched mode: many independent cells resident on the device, whole DoSchedule() loops
 * (EWMA update -> RBsAllocation -> DoStopSchedule accounting) run in one launch.
 * Replaces, per cell and per TTI: RadioBearer::UpdateAverageTransmissionRate
 * (ref: src/flows/radio-bearer.cpp:139-164), SelectSliceToServe (NVS), RBsAllocation,
 * DoStopSchedule's counters (ref: downlink-transport-scheduler.cpp:170-221), the CQI refresh
 * (ref: src/protocolStack/mac/enb-mac-entity.cc:160-193, src/device/CqiManager/cqi-manager.cpp:94-123)
 * and the simulator clock t_k = fl(t_{k-1} + 0.001) (ref: src/core/eventScheduler/simulator.cc:117-126).
 * ------------------------------------------------------------------------------------------ */
typedef struct rs_batch rs_batch;

typedef struct rs_batch_config {
  rs_config cell;            /* every cell of the batch shares this configuration              */
  int32_t n_cells;
  int32_t first_tti;         /* TTI index of the first scheduled TTI (reference: 100 = 0.1 s)   */
  int32_t cqi_refresh;       /* synthetic/epoch source: new grid every this many TTIs (40)      */
  int32_t phy_error_draws;   /* 1: consume one rand() per UE served in the previous TTI, as the
                                reference's PHY error model does on the shared libc stream
                                (ref: src/phy/wideband-cqi-eesm-error-model.cpp:69)              */
  int32_t threads_per_cell;  /* workgroup size, multiple of 64 in [64,512] ([64,1024] with jit = 1: the built-in kernels
                                are bounded at 512); 0 = default: 512, or 256 once the batch puts 4 or more cells on
                                every CU.  More than 512 threads work with jit = 1 but are never chosen automatically
                                (640 on the 64-RBG grid: 8.55 against 13.58 M TTIs/s, profiles/r04_r64.md)          */
  int32_t jit;               /* 1: compile the cell kernel for this batch's exact shape at create time
                                (hiprtc, ~2 s, cached per process); results are identical, the built-in
                                kernels are used if the compilation fails (rs_batch_jit_status tells).  0: built-in kernels.
                                The environment variable RS_JIT=0|1 overrides.  Unlogged launches of at least 256 TTIs
                                (RS_JIT_LEAN_MIN_TTIS) on epoch grids without per-PRB twins, error-model draws or synthetic-experiment
                                blocks run the LEAN build of that kernel (those launch options as compile-time constants; compiled at
                                the first such launch, ~2 s; identical results; RS_JIT_LEAN=0 keeps the general build).  A batch with
                                cqi_refresh <= 4 gets kernels that fetch the next grid during the serial end of the TTI.        */
  int32_t cqi_epoch_wrap;    /* epoch sources (upload / synthesize): 0 = running past the last epoch is RS_ERR_RANGE;
                                1 = the epochs cycle (epoch index modulo n_epochs) -- a bounded set of grids serves a
                                run of any length, e.g. the streamed-CQI measurement with cqi_refresh = 1 (ABI 9)  */
  int32_t queue_state_lds;   /* queue model: where the bearers' hot words (156 B per user) live during a launch.
                                0 = automatic: LDS when the cell still fits the CU's 160 KB, except when that is what
                                takes the cell over 80 KB (one cell per CU instead of two) in a batch of more cells
                                than CUs; 1 = LDS whenever it fits; -1 = HBM (ABI 9)                              */
  int32_t autotune;          /* 1 (with jit = 1; schedulers 8, 9, 101, 103 without the queue model): before the batch's first long
                                unlogged launch (or in rs_batch_prepare_launch) the lean kernel is built in two or three variants
                                the library's rule table chooses between -- LLVM scheduler strategy, speculative next-TTI scan /
                                held winners on or off, users per stage-1 block --, each is timed on the batch's next
                                min(n_ttis, 512) TTIs from a snapshot of the whole cell state that is put back afterwards, and the
                                fastest serves the batch.  Every variant is exact, so results do not depend on the choice; cost
                                ~2 s of hiprtc per variant once per shape and machine (the code objects are cached on disk).
                                rs_batch_autotune_report tells what was measured.  0 = the rule table alone.  Every trial's final
                                state is compared with the rule table's build: a variant that disagrees is never kept.  (ABI 10) */
  int32_t selfcheck;         /* The self-check of a batch's run-time builds (jit = 1, up to 512 threads per cell): before the batch's first
                                launch (or in rs_batch_prepare_launch) its next min(n_ttis, 256) TTIs run on the kernels built into the
                                library and on its run-time compiled ones (general build; the lean build before the first launch that
                                uses it), each from the same snapshot of the whole cell state -- bearers' queues included --, which is put
                                back; the final states must agree bit for bit.  The built-in kernels are one binary -- the one the GPU
                                parity suite checks against the oracle --, a run-time build is a fresh compilation for this shape.  A
                                general build that disagrees is dropped: the batch runs on the built-in kernels and rs_batch_jit_status
                                returns -2 with the reason; a lean build that disagrees alone is dropped alone.
                                0 (default, ABI 11): every build that does not carry the self-check mark -- compiled by this process, or
                                loaded from a cache file no process has checked yet; a build that passes gets the mark in its cache file,
                                so only the first process of a campaign pays (RS_JIT_SELFCHECK=0 switches this default off).
                                1: every build, marked or not.  -1: never.  (ABI 10; default on and -1 since ABI 11)            */
} rs_batch_config;

rs_batch* rs_batch_create(const rs_batch_config* cfg);
rs_batch* rs_batch_create_checked(const rs_batch_config* cfg, int abi_version, size_t cfg_size);
#define RS_BATCH_CREATE(cfg) rs_batch_create_checked((cfg), RS_ABI_VERSION, sizeof(rs_batch_config))
void rs_batch_destroy(rs_batch* b);

/* per-cell libc-compatible rand() streams: srand(seed[c]) then rand_skip[c] values discarded
 * (ref: src/scenarios/single-cell-with-interference.h:84-89, src/utility/seed.h:25-35) */
int rs_batch_seed(rs_batch* b, const uint32_t* seed /* [n_cells] */, const int64_t* rand_skip /* [n_cells] or NULL */);

/* CQI source A: explicit grids. h_cqi = [n_cells][n_epochs][U][R] (host); epoch e serves scheduled
 * TTIs [e*cqi_refresh, (e+1)*cqi_refresh).  Copied to HBM once. */
int rs_batch_upload_cqi_epochs(rs_batch* b, const uint8_t* h_cqi, int32_t n_epochs);
/* CQI source A, per PRB: h_cqi_prb = [n_cells][n_epochs][U][R*rbg_size], reports that may differ inside an RBG (what
 * enb-mac-entity.cc:173-186 stores: all PRBs).  The metric reads PRB rbg*rbg_size (ref: downlink-transport-scheduler.cpp:536), link
 * adaptation every allocated PRB (ref: :643-646). */
int rs_batch_upload_cqi_epochs_prb(rs_batch* b, const uint8_t* h_cqi_prb, int32_t n_epochs);
/* CQI source B: i.i.d. grids drawn on the device from a CQI histogram (weights of CQI 1..15),
 * counter-based generator keyed by (seed, cell, epoch, user, rbg).  Stays in HBM. */
int rs_batch_synthesize_cqi(rs_batch* b, uint64_t seed, const double* cqi_weights /* [15] */, int32_t n_epochs);
/* same with the batch's cells numbered first_cell .. first_cell + n_cells - 1: the generator is keyed by the GLOBAL cell id,
 * so a cell's grids do not depend on how a job's cells are sharded over ranks (rs_batch_synthesize_cqi: first_cell = 0) */
int rs_batch_synthesize_cqi_at(rs_batch* b, uint64_t seed, const double* cqi_weights /* [15] */, int32_t n_epochs,
                               int64_t first_cell);
/* read back the grids of one cell (either source) for the parity tests: [n_epochs][U][R] */
int rs_batch_download_cqi_epochs(rs_batch* b, int32_t cell, uint8_t* h_cqi);
/* CQI source C: the reference's trace replay.  h_trace = [n_traces][n_rows][R]; user u of cell c
 * replays trace h_user_trace[c*U+u]; refresh rule and row index exactly as
 * cqi-manager.cpp:105-123 / enb-mac-entity.cc:189-191 ((int)(Now*1000/40) % row_modulus). */
int rs_batch_set_trace(rs_batch* b, const uint8_t* h_trace, int32_t n_traces, int32_t n_rows,
                       int32_t row_modulus, const int32_t* h_user_trace /* [n_cells][U] */);

/* CQI source C, per PRB: h_trace_prb = [n_traces][n_rows][R*rbg_size] (rs_trace_read_ue_log's out_prb rows); same replay rule */
int rs_batch_set_trace_prb(rs_batch* b, const uint8_t* h_trace_prb, int32_t n_traces, int32_t n_rows, int32_t row_modulus,
                           const int32_t* h_user_trace /* [n_cells][U] */);

/* ---- the reference's CQI trace files (host side of source C; no GPU involved) ----------------------
 * mapping<i>.config: lines "<user id> <trace id>"; user u replays trace map[u % n_entries]
 *   (ref: src/protocolStack/mac/enb-mac-entity.cc:47-53 and :171).  Returns the number of entries read
 *   (at most max_entries are stored), or a negative RS_ERR_*.
 * ue<trace id>.log: one text line per 40-TTI report, nb_rbs space-separated CQI values per line taken from the
 *   start of the line (ref: :173-186, MAX_TTI_TRACE = 475 lines).  rs_trace_read_ue_log parses the first n_rows
 *   lines; out_prb (optional) = [n_rows][nb_rbs]; out_rbg (optional) = [n_rows][nb_rbs/rbg_size], the value of the
 *   first PRB of every RBG -- what the metric reads (downlink-transport-scheduler.cpp:536).  Returns the number of
 *   RBGs whose PRBs do NOT all carry the same value (0 for the shipped traces: then the RBG-granular replay of
 *   rs_batch_set_trace is exact, otherwise use the per-PRB drop-in path), or a negative RS_ERR_*.
 *   A line with fewer than nb_rbs values repeats the last value read, like the reference's extraction loop.
 * rs_trace_load_dir reads ue0.log .. ue<n_traces-1>.log of one directory into out_rbg =
 *   [n_traces][n_rows][nb_rbs/rbg_size] (the h_trace argument of rs_batch_set_trace); same return value. */
/* LDS bytes one cell occupies (the kernel's carve for this shape; no GPU needed).  160 KB per CU: <= 40 960 B keeps four
 * cells on a CU, <= 81 920 B two. */
int rs_lds_bytes_per_cell(int n_slices, int n_users, int n_rbgs, int sched, int threads);

/* measurement helper: streaming copy of `bytes` (16 B per lane), `iters` launches timed with HIP events;
 * *copy_gbs = (bytes read + bytes written) / time.  Quoted by bench.py next to the 8 TB/s HBM3E spec. */
int rs_hbm_copy_probe(int device, uint64_t bytes, int iters, double* copy_gbs);

int rs_trace_read_mapping(const char* path, int32_t* trace_of_entry, int32_t max_entries);
int rs_trace_read_ue_log(const char* path, int32_t n_rows, int32_t nb_rbs, int32_t rbg_size, uint8_t* out_rbg,
                         uint8_t* out_prb);
int rs_trace_load_dir(const char* dir, int32_t n_traces, int32_t n_rows, int32_t nb_rbs, int32_t rbg_size,
                      uint8_t* out_rbg);

/* ------------------------------------------------------------------------------------------
 * Finite queues in a batch: what customised slices (algo_alpha = 1) and rate-limited traffic need.
 * Replaces, per cell: the bearers' MAC queues and RLC dequeue (ref: src/flows/MacQueue.cpp:86-200,
 * src/protocolStack/rlc/um-rlc-entity.cpp:126-196), HasPackets / GetQueueSize / GetHeadOfLinePacketDelay
 * (src/flows/radio-bearer.cpp:263-367), SelectFlowsToSchedule + InsertFlowToUser (downlink-transport-scheduler.cpp:105-150,
 * packet-scheduler.cpp:305-335: users without queued data are not scheduled, slice_priority_, dataToTransmit) and
 * DoStopSchedule's split of a grant over the user's bearers from the highest priority down (:170-221).
 * Schedulers 8, 9, 101, 103 and, RBG by RBG on one wave because their gates bind with finite queues, 7 (a user leaves the race
 * once its PRBs reach m_requiredRBs, downlink-nvs-scheduler.cpp:299-300 with packet-scheduler.cpp:319-334) and 1 (every bearer
 * with packets is a flow with its own PF average; a flow leaves once the transport block of its PRBs so far carries its queue,
 * downlink-packet-scheduler.cpp:221-265; rbg_to_user then holds flow ids 2 * user + bearer and DoStopSchedule credits whole
 * transport blocks, dl-pf-packet-scheduler.cpp:60-125).  The applications themselves stay outside: the caller hands in every
 * bearer's arrival bursts.
 * Restriction: a batch's band is exactly n_rbgs * rbg_size PRBs.  The reference's wideband CQI behind m_requiredRBs
 * (packet-scheduler.cpp:321-327) runs over cqiFeedbacks.size() = every PRB of the band, including the nb_rbs % rbg_size
 * trailing PRBs that are never allocated (25 PRBs with RBGs of 2, 50 with 3, 75 with 4); a batch has no such PRBs, so scheduler 7
 * with queues matches the reference on bands whose PRB count is a multiple of the RBG size -- every shipped configuration
 * (100 MHz: 512 PRBs in RBGs of 8; the trace grid).
 * Round 3: a launch keeps the bearers' hot words in LDS (156 B per user) when the cell still fits the CU's 160 KB, else in HBM.
 * ------------------------------------------------------------------------------------------ */
/* bearer_kind [U][2], index = bearer priority (RadioBearer::GetPriority, radio-bearer.cpp:88-97): 0 none, 1 InfiniteBuffer
 * (always has packets, dataToTransmit 1e8), 2 finite MAC queue.  Every user needs at least one bearer.  Before the first run. */
int rs_batch_set_bearers(rs_batch* b, const uint8_t* bearer_kind);
/* arrival bursts of every finite-queue bearer: bearer (cell c, user u, priority k) owns bursts offsets[(c*U+u)*2+k] ..
 * offsets[(c*U+u)*2+k+1] of the three arrays (ascending in time).  At time[i] the application enqueues n_full[i] packets of
 * 1495 bytes (MAXMTUSIZE 1490 + UDP 8 + IP 20, ROHC 28 -> 3, + PDCP 2) and then, if last_bytes[i] > 0, one of last_bytes[i]
 * bytes; a burst is visible to the TTI whose time stamp is >= time[i]. */
int rs_batch_set_arrivals(rs_batch* b, const int64_t* offsets, const double* time, const int32_t* n_full, const int32_t* last_bytes);
/* per bearer, [n_cells][U][2]: PF average, cumulative bytes / RBs (the `cumu_bytes:` / `cumu_rbs:` of the bearer's log lines),
 * MAC queue bytes and packets; any pointer may be NULL */
int rs_batch_read_bearer_state(rs_batch* b, double* avg_rate, int64_t* cum_bytes, int64_t* cum_rbs, int32_t* queue_bytes,
                               int32_t* queue_packets);
/* host-side generator (no GPU): the arrival bursts of one InternetFlow application (ref: src/flows/application/
 * InternetFlow.cpp: exponential inter-arrival times rounded up to ms, heavy-tailed flow sizes, 1490-byte packets) between
 * start_time and stop_time; flow sizes from a libc-compatible rand() stream seeded with size_seed.  Returns the number of
 * bursts written (<= max_bursts) or a negative RS_ERR_*. */
int rs_internet_flow_arrivals(double rate_mbps, double start_time, double stop_time, uint32_t size_seed, int32_t max_bursts,
                              double* time, int32_t* n_full, int32_t* last_bytes);

/* run n_ttis scheduled TTIs of every cell in ONE kernel launch on the batch's stream and wait */
int rs_batch_run(rs_batch* b, int32_t n_ttis);
/* same, not waiting (for overlap and hipEvent timing by the caller) */
int rs_batch_run_async(rs_batch* b, int32_t n_ttis);
int rs_batch_sync(rs_batch* b);
/* same as rs_batch_run, also returning the per-TTI decisions of every cell (parity tests, log writer):
 * h_rbg_to_user [n_cells][n_ttis][R] (int16, -1 = none), h_tbs_bits [n_cells][n_ttis][U] (int32),
 * h_quota / h_target [n_cells][n_ttis][S] (int16), h_uinfo [n_cells][n_ttis][U] (int32:
 * nPRB | final_cqi << 16 | mcs << 24, 0 = not scheduled); any may be NULL */
int rs_batch_run_logged(rs_batch* b, int32_t n_ttis, int16_t* h_rbg_to_user, int32_t* h_tbs_bits,
                        int16_t* h_quota, int16_t* h_target, int32_t* h_uinfo);
/* the same with every log optional in one struct, plus what the inter-slice step reads (parity tests pin
 * GreedyByRow / MaximizeCell / Vogel on the device's own per-TTI inputs with it) */
typedef struct rs_batch_log {
  int16_t* rbg_to_user;  /* [n_cells][n_ttis][R] */
  int32_t* tbs_bits;     /* [n_cells][n_ttis][U] */
  int16_t* quota;        /* [n_cells][n_ttis][S] */
  int16_t* target;       /* [n_cells][n_ttis][S] */
  int32_t* uinfo;        /* [n_cells][n_ttis][U] */
  uint32_t* slice_keys;  /* [n_cells][n_ttis][R][S], RS_SCHED_SEQUENTIAL / MAXCELL / UPPERBOUND / SUBOPT / VOGEL only:
                            CQI of the slice's best user on the RBG (0: the slice has no user) | (user id + 1) << 8, i.e.
                            flow_spectraleff / user_index of ref :545-567 */
} rs_batch_log;
int rs_batch_run_logged_ex(rs_batch* b, int32_t n_ttis, const rs_batch_log* log);
/* `launches` back-to-back launches of n_ttis each, timed with HIP events on the batch's stream;
 * ms_per_launch[launches] receives each launch's duration */
int rs_batch_run_timed(rs_batch* b, int32_t n_ttis, int32_t launches, float* ms_per_launch);

/* per-bearer state after the runs so far; any pointer may be NULL.
 * avg_rate [n_cells][U], cum_bytes/cum_rbs [n_cells][U] (RadioBearer::GetCumulateBytes/RBs),
 * slice_state [n_cells][S] (slice_rbs_offset_, or slice_ewma_time_ for NVS) */
int rs_batch_read_state(rs_batch* b, double* avg_rate, int64_t* cum_bytes, int64_t* cum_rbs,
                        double* slice_state);
/* An initialisation / test hook, NOT a checkpoint: sets avg_rate [n_cells][U] (every value >= 1, the EWMA's clamp) and
 * slice_state [n_cells][S]; either may be NULL = left alone.  It does not touch the rest of a cell's state -- the grant of the last
 * TTI that the next EWMA update consumes, the clock, the rand() ring, the cumulative counters -- so it is exact before the first
 * launch (or on a batch whose last grant was zero); after a launch the pending grant is applied on top of the new averages.  Not
 * with the queue model (one average per bearer there).  (ABI 9) */
int rs_batch_write_state(rs_batch* b, const double* avg_rate, const double* slice_state);
/* Checkpoint / resume (ABI 10).  A checkpoint is everything a batch carries from one launch to the next -- PF averages, the grants
 * the next EWMA update consumes, cumulative counters, slice state, every cell's clock, rand() ring and CQI-report state, the number of
 * TTIs done, and with the queue model the bearers' queues, averages and counters -- as one host block of rs_batch_checkpoint_bytes
 * bytes.  rs_batch_checkpoint_load puts it into a batch of the same shape (slices, UEs, RBGs, PRBs per RBG, scheduler, cells, queue
 * model or not; jit, threads_per_cell, device and process may differ), which then continues bit for bit like the batch that saved
 * it: run 300 TTIs, save, run 300 more == load into a new batch, run 300.  The CQI source (epoch grids, trace tables, arrival bursts) is
 * configuration, not state: set it on the resuming batch as on the first one (rs_batch_seed need not be called: the ring is in the
 * block).  Between launches only.  rs_batch_read_state / rs_batch_write_state remain the partial accessors they were. */
int64_t rs_batch_checkpoint_bytes(rs_batch* b);
int rs_batch_checkpoint_save(rs_batch* b, void* buf, size_t buflen);
int rs_batch_checkpoint_load(rs_batch* b, const void* buf, size_t buflen);
/* the simulated clock of every cell (ref: src/core/eventScheduler/simulator.cc:117-126): t [n_cells] = time stamp of the next
 * TTI, last_update [n_cells] = RadioBearer::m_lastUpdate; either may be NULL */
int rs_batch_read_clock(rs_batch* b, double* t, double* last_update);
/* 1: the shape-specialised (hiprtc) kernel is in use (msg is empty, says that the untuned variant had to be built, or reports a
 * passed rs_batch_config.selfcheck); 0: it was not asked for; -1: it was asked for and could not be built -- the built-in kernels
 * run instead and msg receives the reason; -2: it was built and rs_batch_config.selfcheck found its results different from the
 * built-in kernels': dropped, the built-in kernels run */
int rs_batch_jit_status(rs_batch* b, char* msg, size_t msglen);
/* what rs_batch_config.autotune measured: one line "variant: ms" per candidate and the one kept; empty before the tuning ran or
 * when it does not apply.  Returns the number of candidates timed (0: none). */
int rs_batch_autotune_report(rs_batch* b, char* msg, size_t msglen);
/* Optional: build now whatever kernel an unlogged rs_batch_run of n_ttis TTIs would otherwise build at its first launch (the lean
 * build of the shape-specialised kernel, see rs_batch_config.jit), so that no timed launch carries a hiprtc run.  Call it once the
 * CQI source is set.  RS_OK also when there is nothing to build. */
int rs_batch_prepare_launch(rs_batch* b, int32_t n_ttis);
/* per-slice cumulative bytes summed over the batch's cells, reduced on the device into
 * d_out[S] (device pointer, uint64) on the batch's stream -- the vector the multi-GPU run
 * all-reduces over RCCL (the reference's plot_throughput.py:26-56 sums it per slice post hoc) */
int rs_batch_slice_bytes_device(rs_batch* b, uint64_t* d_out);
int rs_batch_slice_bytes(rs_batch* b, uint64_t* h_out /* [S] */);
/* build check (needs no GPU): compile the shape-specialised kernel for one shape with hiprtc; returns the
 * code object size or a negative value with the compiler log in err */
int rs_jit_selfcheck(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, char* err,
                     size_t errlen);
/* the same without the internal -mllvm tuning options: the build the library falls back to (and reports through
 * rs_batch_jit_status: return value 1 with a non-empty message) when hiprtc refuses them */
int rs_jit_selfcheck_untuned(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, char* err,
                             size_t errlen);
/* the same for the queue-model kernel (the code object rs_batch_set_bearers switches a batch to; schedulers 1, 7, 8, 9, 101, 103) */
int rs_jit_selfcheck_queue(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, char* err,
                           size_t errlen);
/* ---- the run-time compiled kernels' cache on disk (ABI 10) ----
 * Every code object hiprtc produces (rs_batch_create with jit = 1, the lean / streamed variants, rs_ctx_specialize) is kept in
 * $RS_JIT_CACHE_DIR (default $XDG_CACHE_HOME/radiosaber_amd, else ~/.cache/radiosaber_amd; created with mode 0700), one file per (device
 * source hash, compiler identity -- see rs_jit_compiler_identity --, full hiprtc option list); a later process loads it instead of
 * compiling (~2 s per variant -> a few ms).  Files are written under a temporary name and renamed; a file that does not check out
 * (magic, key text, length, checksum), is not a regular file of the caller's own, or is writable by anybody else is compiled again and
 * replaced (RS_JIT_CACHE_SHARED=1 accepts another account's read-only cache).  Its last 8 bytes say whether the object has passed a
 * self-check against the built-in kernels ("VERIFIED"); a build that a self-check rejects is unlinked.  RS_JIT_CACHE=0 switches the
 * cache off.
 * rs_jit_cache_stats: this process's hits, misses (= hiprtc runs), files written, files rejected.
 * rs_jit_cache_file: the file the batch kernel of a shape lives in (flags: 2 = streamed-CQI variant, 4 = lean build); returns its
 *   length, 0 when no cache directory can be named.
 * rs_jit_cache_warm: that kernel THROUGH the cache (compiles and stores on a miss); needs no GPU; code size or -1 with the log. */
void rs_jit_cache_stats(long long out[4]);
/* the compiler identity that is part of every cache key (ABI 11): hiprtc and HIP runtime versions down to the patch level, the clang
 * version string of the compiler inside hiprtc with its LLVM commit (read from a two-line probe compilation), comgr's version.
 * RS_JIT_COMPILER_ID replaces it (deployments that pin the toolchain themselves).  Needs no GPU. */
const char* rs_jit_compiler_identity(void);
int rs_jit_cache_file(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, int flags, char* out, size_t outlen);
int rs_jit_cache_warm(int n_slices, int n_users, int n_rbgs, int rbg_size, int threads, int sched, int flags, char* err, size_t errlen);
/* 16 hex digits: FNV-1a hash of the device sources this library was built from (and compiles at run time); measurement
 * records under profiles/ carry it so that a record taken on other kernel code can be told apart (bench.py: "stale") */
const char* rs_device_source_hash(void);
/* diagnostics: how often the std::sort emulation (MaximizeCell, UpperBound) took libstdc++'s heap-sort fallback -- std::__partial_sort,
 * when std::__introsort_loop's depth limit 2 * floor(log2 n) runs out on a range longer than 16 (bits/stl_algo.h:1937-1957; the
 * reference's call: downlink-transport-scheduler.cpp:354-361) -- since the batch / context was created, per device site:
 * [0] workgroup level of the register form, [1] inside a single wave's finish of the last levels, [2] workgroup level of the LDS
 * form (more than four sort records per thread).  Random CQI grids never get there; tests/golden/sort_killers.npz holds grids that
 * do.  out = [n_cells][3] / [3].  (ABI 10) */
int rs_batch_debug_heap_sorts(rs_batch* b, int64_t* out);
int rs_ctx_debug_heap_sorts(rs_ctx* ctx, int64_t* out);
/* diagnostics: the shader clock the LAST launch really ran at, per cell: (s_memtime cycles) / (s_memrealtime ticks at 100 MHz) between
 * the first and the last instruction of the cell's thread 0 -> shader_mhz [n_cells]; kernel_ms [n_cells] = that span in milliseconds
 * (a cell's own run time: cells that waited for a free CU start later).  Either may be NULL.  (ABI 10) */
int rs_batch_debug_clocks(rs_batch* b, double* shader_mhz, double* kernel_ms);
/* diagnostics: cycles per kernel phase of one cell's first thread, summed over the last launch;
 * only in the separate -DRS_STAMPS build (RS_ERR_STATE in the product library) */
int rs_batch_debug_stamps(rs_batch* b, int32_t cell, uint64_t* out20);
/* scheduled TTIs completed per cell so far */
int64_t rs_batch_ttis_done(rs_batch* b);
/* the hipStream_t the batch launches on */
void* rs_batch_stream(rs_batch* b);
/* name of the dominant kernel (for matching rocprofv3 --kernel-trace rows) */
const char* rs_batch_kernel_name(rs_batch* b);

#ifdef __cplusplus
}
#endif
#endif /* RADIOSABER_HIP_H_ */
