/*
 * rs_device.h -- data structures shared by the host side of the C ABI (rs_api.hip) and the
 * gfx950 kernels (rs_kernels.hip).  Not part of the public interface.
 */
#ifndef RS_DEVICE_H_
#define RS_DEVICE_H_

#ifndef __HIPCC_RTC__
#include <stdint.h>
#elif !defined(RS_RTC_STDINT)
#define RS_RTC_STDINT /* hiprtc keeps its fixed-width types in a namespace */
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif

#define RS_WAVE 64
/* tx word of a shape-specialised batch kernel: bytes granted since the last EWMA update | PRBs << 20 | "counted" << 30 */
#define RS_TX_BYTES_MASK 0xFFFFF
#define RS_TX_NPRB_SHIFT 20
#define RS_TX_NPRB_MASK 0x3FF
#define RS_TX_COUNTED (1 << 30)
/* a shape-specialised kernel counts a launch's bytes per user in 32 bits: 32 768 TTIs x < 2^16 bytes per TTI */
#define RS_FULL_PACKET 1495 /* MAXMTUSIZE 1490 + UDP 8 + IP 20, ROHC 28 -> 3, PDCP 2 (ref: src/protocolStack/packet/Packet.cpp:84-118) */
#define RS_MAX_TTIS_PER_LAUNCH 32768
#define RS_MAX_BYTES_PER_TTI 65535
#define RS_PF_SEG 32 /* sched 1: users are scanned in segments of this many for the per-RBG argmax */

/* per-cell scratch in LDS (host needs its size for the LDS carve) */
#define RS_MAX_SEGS 256 /* > 4096/17 sub-ranges longer than 16 on one recursion level */
/* hand-shake words of one serial phase (speculative next-TTI scan, rs_kernels.hip): waves that finished the speculative EWMA /
 * the speculative scan, "the allocation is decided", and the number of items whose winner was served */
struct RsSpecFlags {
  int32_t ctr_p1, ctr_p3, greedy_done, n_fix;
};
#define RS_FIX_CAP 960 /* u16 entries of RsMisc::hist behind the 128-byte served bitmap */
struct RsMisc {
  unsigned long long maskA[64], maskB[64]; /* level-synchronous introsort: stop ballots per chunk */
  int32_t seg_begin[68];
  int32_t target[64];
  int32_t quota[64];
  int32_t n_level[48];           /* level-synchronous introsort: live sub-ranges per recursion level */
  uint16_t hist[64 * 16];        /* counting sort: per 64-element chunk, per key */
  int32_t mcs_of_cqi[16];
  int32_t tbs1_of_cqi[16];       /* TBS bits of ONE PRB at a CQI (the synthetic-experiment transport block) */
  int32_t served;
  int32_t nvs_slice;
  int32_t pad[2];
  float ones16[16];              /* numerator table of an epsilon = 0 slice (pow(x, 0) = 1) */
  double eff16[16];              /* Vogel: flow_spectraleff of a key (key 0 = empty slice = 0.0) */
  uint8_t eps_psi[64];           /* per slice: bit 0 = algo_epsilon, bit 1 = algo_psi (read by every work item of P3) */
  int32_t rcp_off[64];           /* per slice: window start in the reciprocal array minus the slice's 8-aligned first user */
  RsSpecFlags spec[2];           /* by TTI parity */
  int32_t grid_free;             /* TTIs of this launch whose serial wave is done reading the CQI grid (the next grid may be written over it) */
  int32_t heap_sorts[3];         /* diagnostics: std::__partial_sort fallbacks of this launch, per device site (rs_sort_device.h) */
  int32_t prio_boost;            /* this cell is behind the batch's average progress: its waves ask for one issue-priority level more (RS_SETPRIO) */
  int32_t pad3[3];
};

/* LDS carve of one cell (byte offsets from the dynamic LDS base), a pure function of the cell shape so
 * that the host and a shape-specialised kernel build agree on it */
struct RsCarve {
  int Upad, n_seg, n_items, ept, nvs_seg;
  int off_avgk, off_rcp, off_tab, off_slice, off_tx, off_misc, off_tbs, off_elems, off_sorted, off_items,
      off_sortx, off_cqi, off_queue, off_qstate, q_lds, lds_bytes;
};
constexpr int rs_round_up(int x, int a) { return (x + a - 1) / a * a; }
constexpr int rs_upad_of(int U) {
  /* 8 * odd: 8-byte column reads of 32 consecutive RBGs hit 32 distinct bank pairs */
  int k = (U + 7) / 8;
  return 8 * ((k & 1) ? k : k + 1);
}
/* sched 11 (NVS non-greedy sampler) scratch, at off_sortx: val f64[U][4] (a slice of up to 64 users: its 4 n metrics, then the
 * u16 key table; rs_nvs_val_bytes) | draws u8[draw bytes] | winner u16[batch][R rounded up to 4] (metric index i * 4 + draw, 0x8000 = the metric is 0.0) | pad |
 * high u8[U].  Round 5: a sample's RBG metrics are no longer stored as doubles (f64[batch][R], 16 ... 32 KB at 64 RBGs) -- the
 * lane that adds them up reads them through the winner's metric index; with a 4 KB draw buffer that is two 500-UE x 64-RBG cells
 * per CU instead of one. */
#ifndef RS_UMAP_SCRATCH_BYTES
#define RS_UMAP_SCRATCH_BYTES 272 /* sched 101 (SubOpt), at off_sortx: rs_umap_order's nxt u8[68] | bkt u8[128] | ord u8[64] */
#endif
#define RS_NVS_SAMPLES 300      /* num_sample, downlink-nvs-scheduler.cpp:430 */
#define RS_NVS_DRAW_BYTES 8192  /* draws of two batches of samples (one being drawn while the other is scanned); 4096 where that keeps two cells per CU */
#define RS_NVS_BATCH 64         /* samples per batch at most (one lane of wave 0 per sample adds up its RBGs); rs_carve picks 64 or 32 */
/* RsCarve::nvs_seg of scheduler 11 carries the three choices: samples per batch | draw-buffer KB << 8 | metric / key array in 16 B << 16 */
constexpr int rs_nvs_pack(int batch, int draw_bytes, int val_bytes) { return batch | ((draw_bytes >> 10) << 8) | ((val_bytes >> 4) << 16); }
constexpr int rs_nvs_batch_of(int seg) { return seg & 0xff; }
constexpr int rs_nvs_draw_of(int seg) { return ((seg >> 8) & 0xff) << 10; }
constexpr int rs_nvs_val_of(int seg) { return (seg >> 16) << 4; }
/* a slice's 4 n metrics + its key table u16[n][4][Q][4]: a row per (UE, draw), the RBGs in Q = ceil(R / 4) units of four */
constexpr int rs_nvs_key_bytes(int n, int R) { return 32 * n + 32 * n * ((R + 3) / 4); }
/* the metric array: f64[4] per user of the served slice, and behind the metrics of a slice of up to 64 users its key table.  Sized for
 * the batch's longest slice (win: its 8-aligned window; U when that is not known) with its keys -- a cell of few slices scans on keys
 * too -- and for U users on doubles (32 U: what a ragged batch's shorter slices may use for their keys when the longest has more than
 * 64 users), whichever is larger; where that costs the second cell on the CU, the smaller of the two that still serves the longest
 * slice, then 32 U (the window overstates the slice by up to 14 users: 500 UEs x 64 RBGs in slices of 25) (round 5) */
constexpr int rs_nvs_val_bytes(int U, int R, int win, int choice) { /* 0: the larger of the two, 1: the longest slice with its keys, 2: 32 U */
  const int w = win > 0 && win < U ? win : U;
  const int keyed = w <= 64 ? rs_nvs_key_bytes(w, R) : 0;
  const int longest = keyed > 32 * w ? keyed : 32 * w;
  return rs_round_up(choice == 1 ? longest : (choice == 2 ? 32 * U : (longest > 32 * U ? longest : 32 * U)), 16);
}
constexpr int rs_nvs_scratch_bytes(int U, int R, int seg) {
  return rs_nvs_val_of(seg) + rs_nvs_draw_of(seg) + 2 * rs_nvs_batch_of(seg) * ((R + 3) / 4 * 4) + 128 + (U + 15) / 16 * 16;
}
/* sched 7: the served slice is scanned in 8-aligned runs of nvs_seg users, one work item per (run, RBG); the run winners
 * (user u16 + metric f64 per RBG) are reduced per RBG in ascending order afterwards.  With slices of more than 32 users on
 * average rs_carve picks the smallest run length in {8, 16, 32} that keeps the cell at or under 80 KB of LDS (two cells per
 * CU); otherwise nvs_seg = 0: one work item per RBG scans the whole slice (measured on 25-user slices: runs cost 8-10 %). */
/* queue != 0: schedulers 1 and 7 allocate RBG by RBG on one wave (the per-flow "satisfied" break / the m_requiredRBs gate) and
 * keep per-bearer scratch in LDS: grant1 i32[U] | data0 i32[U] | data1 i32[U] | need i32[U] | flags u8[2U]  (queue = 1: only
 * that -- the drop-in contexts' gate scratch).
 * queue = 3: the queue model with the bearers' words in HBM (the host's choice when LDS residency would halve the cells per CU).
 * queue = 2: the batch runs the queue model (finite MAC queues, two bearers per user) and, when the cell still fits the CU's
 * 160 KB, keeps the bearers' hot words in LDS for the whole launch instead of reading and writing them in HBM every TTI
 * (RS_QSTATE_BYTES_PER_USER per user at off_qstate, q_lds = 1; round 3, profiles/r03_queue_mode.md):
 *   avg f64[2][U] | next arrival time f64[2][U] | head burst time f64[2][U] | HoL delay f64[U] |
 *   head, tail, pk, frag, bytes, pkts, tx i32[7][2][U] | first burst i64[2][U] | bursts, head burst's n_full / last i32[3][2][U] |
 *   bearer kind u8[U][2] | user flags u8[U] | user slice u8[U] */
#define RS_QSTATE_BYTES_PER_USER 156
#define RS_LDS_LIMIT (160 * 1024)
/* Register form of the sort's workgroup levels (rs_sort_device.h): one cut slot per 16 positions, and -- round 6 -- a 16-bit stop-rank
 * entry per position.  Up to 1 024 positions the ranks live in RsMisc::hist (free between the metric scan and the counting sort: one,
 * two positions per thread keep their LDS footprint -- four 500 x 25 cells per CU); longer arrays (the 64-RBG grid: 1 280) get their own
 * room behind the cut slots. */
constexpr int rs_sort_cuts_bytes(int N) { return rs_round_up(4 * (N / 16 + 2), 16); }
constexpr int rs_sort_ranks_bytes(int N) { return ((N + 63) / 64) * 64 > 1024 ? 2 * 64 * ((N + 63) / 64) : 0; }
constexpr RsCarve rs_carve_with(int S, int U, int R, int sched, int threads, int nvs_seg, int queue = 0) {
  RsCarve c{};
  c.Upad = rs_upad_of(U);
  c.nvs_seg = nvs_seg;
  c.n_seg = sched == 1 ? (U + RS_PF_SEG - 1) / RS_PF_SEG
                     : (sched == 7 ? (nvs_seg ? (U + nvs_seg - 1) / nvs_seg + 1 : 1) : (sched == 11 ? 1 : S));
  c.n_items = R * c.n_seg;
  c.ept = (R * S + threads - 1) / threads;
  const bool pf_like = sched == 1 || (sched == 7 && nvs_seg != 0); /* winner tables instead of sort records */
  int off = 8 * U; /* avg */
  c.off_avgk = off; off += 8 * U;
  /* stage-1 reciprocals, one 8-aligned window per slice with zeros around it (f32[Upad + 16 S]), then each user's
   * window offset (i16[U]) */
  c.off_rcp = off; off += rs_round_up(4 * (c.Upad + 16 * S) + 2 * U, 16);
  c.off_tab = off; off += 8 * 48 + 64;
  c.off_slice = off; off += 8 * 128;
  c.off_tx = off; off += rs_round_up(4 * U, 16);
  c.off_misc = off; off += rs_round_up((int)sizeof(RsMisc), 16);
  c.off_tbs = off; off += rs_round_up(4 * 16 * (R + 1), 16); /* TBS bits of n RBGs at a final CQI */
  c.off_elems = off; off += sched == 11 ? 0 : rs_round_up(pf_like ? 8 * c.n_items : 4 * R * S, 16); /* (the sampler keeps no records) */
  /* (the sorted / alternate record array: the transport schedulers only -- 1, 7 and 11 keep winner tables; without it a 600-UE x 64-RBG
   * cell of the per-flow PF scheduler is 77 824 B instead of 82 944: two cells per CU, round 5) */
  c.off_sorted = off; off += (sched == 1 || sched == 7 || sched == 11) ? 0 : rs_round_up(4 * R * S, 16);
  /* winners per work item; the schedulers with a speculative next-TTI scan keep two TTIs' worth (by parity) */
  const bool spec = sched == 8 || sched == 9 || sched == 101 || sched == 103;
  c.off_items = off; off += rs_round_up((spec ? 4 : 2) * c.n_items, 16);
  /* level-synchronous introsort scratch: cut per sub-range (+ bounds/pivots when the state lives in LDS) */
  /* register form (ept <= 4): one cut slot per 16 positions; LDS form: bounds, pivots and cuts per position */
  c.off_sortx = off; off += (sched == 9 || sched == 10) ? rs_round_up(c.ept <= 4 ? rs_sort_cuts_bytes(R * S) + rs_sort_ranks_bytes(R * S) : 8 * R * S, 16)
                                                           : (sched == 11 ? rs_round_up(rs_nvs_scratch_bytes(U, R, nvs_seg), 16) : (sched == 101 ? RS_UMAP_SCRATCH_BYTES : 0));
  c.off_cqi = off; off += rs_round_up(c.Upad * R, 16);
  c.off_queue = off; off += (queue && (sched == 1 || sched == 7)) ? rs_round_up(18 * U, 16) : 0;
  c.off_qstate = off;
  c.q_lds = (queue == 2 && off + rs_round_up(RS_QSTATE_BYTES_PER_USER * U, 16) <= RS_LDS_LIMIT) ? 1 : 0;
  off += c.q_lds ? rs_round_up(RS_QSTATE_BYTES_PER_USER * U, 16) : 0;
  c.lds_bytes = off;
  return c;
}
#ifndef RS_NVS_WHOLE_SLICE
#define RS_NVS_WHOLE_SLICE 64 /* sched 7: up to this many users per slice the slice is scanned whole -- the shape-specialised kernels scan it
                                * four lanes per item during the previous TTI's serial phase (kQuad7, windows of up to 64 users): 1 000 UEs in
                                * 20 slices 215.1 against 179.9 M TTIs/s with split runs, 106.4 against 85.3 at 64 RBGs */
#endif
#ifndef RS_NVS_WHOLE_SLICE_BUILTIN
#define RS_NVS_WHOLE_SLICE_BUILTIN 32 /* ... the kernels built into the library have no such scan: split runs from 32 users per slice on
                                        * (whole slices lose 6 % at 50 users per slice there: 88.9 against 94.2 M TTIs/s, tools/experiments/r04/run45.sh) */
#endif
/* win: the batch's longest 8-aligned slice window when the caller knows it (batches: slice_window(); RS_JIT_WIN in their kernels), 0 = not
 * known (drop-in contexts, the build checks): the NVS rule is keyed on it, so that ONE long slice of a ragged batch gets split runs
 * (round 5; the average was the key before, and such a batch got neither split runs nor the four-lane scan).  nvs_whole: the
 * threshold of that rule -- RS_NVS_WHOLE_SLICE for a batch that asked for its shape-specialised kernel, _BUILTIN otherwise. */
constexpr RsCarve rs_carve(int S, int U, int R, int sched, int threads, int queue = 0, int win = 0, int nvs_whole = RS_NVS_WHOLE_SLICE) {
  if (queue) return rs_carve_with(S, U, R, sched, threads, 0, queue); /* (sched 7 with queues: no split runs, no metric scan) */
  const bool nvs_long = sched == 7 && (win > 0 ? win > nvs_whole : U > nvs_whole * S);
  if (nvs_long) { /* a slice of more users than that: one run per work item is too long */
    for (int seg = 8; seg <= 16; seg *= 2) {
      const RsCarve c = rs_carve_with(S, U, R, sched, threads, seg);
      if (c.lds_bytes <= 80 * 1024) return c;
    }
  }
  if (nvs_long) return rs_carve_with(S, U, R, sched, threads, 32);
  if (sched == 11) {
    /* the sampler's batch and draw buffer (carried in nvs_seg, which scheduler 7 alone reads as a run length): 64 samples -- half
     * the barriers, fuller scan rounds: 15.3 against 13.0 M TTIs/s at 500 UEs x 25 RBGs -- and 8 KB of draws, unless a smaller choice
     * is what keeps the cell within 80 KB (two cells per CU: 100 UEs x 64 RBGs 12.1 M with 32 against 7.7 M with 64) */
    for (int choice = 0; choice <= 2; ++choice) { /* the roomiest metric / key array that keeps two cells on a CU */
      const int val = rs_nvs_val_bytes(U, R, win, choice);
      const RsCarve c0 = rs_carve_with(S, U, R, sched, threads, rs_nvs_pack(64, RS_NVS_DRAW_BYTES, val));
      if (c0.lds_bytes <= 80 * 1024) return c0;
      const RsCarve c1 = rs_carve_with(S, U, R, sched, threads, rs_nvs_pack(64, RS_NVS_DRAW_BYTES / 2, val));
      if (c1.lds_bytes <= 80 * 1024) return c1;
      const RsCarve c2 = rs_carve_with(S, U, R, sched, threads, rs_nvs_pack(32, RS_NVS_DRAW_BYTES / 2, val));
      if (c2.lds_bytes <= 80 * 1024) return c2;
    }
    return rs_carve_with(S, U, R, sched, threads, rs_nvs_pack(64, RS_NVS_DRAW_BYTES, rs_nvs_val_bytes(U, R, win, 0)));
  }
  return rs_carve_with(S, U, R, sched, threads, 0); /* sched 7 with small slices: one work item per RBG scans the whole slice */
}

/* link-adaptation constants (host libm -> device), see rs_link_tables() in radiosaber_hip.h */
struct RsTables {
  double kbps[16];   /* metric numerator of sched 7/8/9: eff*180000/1000              */
  double pfnum[16];  /* metric numerator of sched 1: eff*180000.                      */
  double eff[16];    /* spectral efficiency of a CQI (0 for "no user"): VogelApproximate's differences */
  double eesm_e[16]; /* E[c] = exp(-10^(SINR[c]/10))                                  */
  double eesm_x[16]; /* decision thresholds X[1..13]                                  */
  int32_t mcs_of_cqi[16];
  int32_t itbs_of_cqi[16];
  int32_t tbs_row_m1[28]; /* the reference's T[-1][itbs] at -O0 (see rs_kernels.hip)  */
  int32_t tbs1_of_cqi[16]; /* TBS bits of one PRB at a CQI (GetTBSizeFromMCS(mcs), AMCModule.cpp:299-303): m_requiredRBs */
  int32_t tbs1_syn[16];    /* the synthetic-experiment build's bits of one PRB reported at a CQI:
                            * GetTBSizeFromMCS(GetMCSFromCQI(GetCQIFromSinr(GetSinrFromCQI(cqi))), 1), downlink-transport-scheduler.cpp:656-658 */
};

/* per-cell scalar state that survives between launches */
struct RsCellScalars {
  double t;            /* simulated time of the next TTI                               */
  double last_update;  /* RadioBearer::m_lastUpdate (same for every bearer of the cell) */
  int64_t last_sent;   /* CqiManager::m_lastSent                                       */
  int32_t reported;    /* first CQI report done                                        */
  int32_t served_prev; /* UEs served in the previous TTI                               */
  int64_t n_done;      /* scheduled TTIs so far                                        */
  int32_t rng_f, rng_b;
  uint32_t rng_r[32];  /* glibc TYPE_3 ring (31 words used)                            */
  int32_t cqi_row;     /* trace row of the last CQI report (reloaded when a launch starts between reports) */
  int32_t pad_;
  int64_t heap_sorts[3]; /* diagnostics: heap-sort fallbacks of the sort emulation so far, per device site (rs_batch_debug_heap_sorts) */
  /* diagnostics: the last launch's first and last instruction of this cell's thread 0 on the shader clock (s_memtime) and on the
   * constant 100 MHz clock (s_memrealtime): their ratio is the shader clock the kernel really ran at (rs_batch_debug_clocks) */
  uint64_t clk_begin, clk_end, real_begin, real_end;
};

enum { RS_CQI_NONE = 0, RS_CQI_EPOCHS = 1, RS_CQI_TRACE = 2 };

/* kernel argument block */
struct RsLaunch {
  /* geometry */
  int32_t S, U, R, G;        /* slices, users, RBGs, PRBs per RBG */
  int32_t Upad;              /* LDS row stride of the RBG-major CQI grid: 8 * odd >= U */
  int32_t nvs_seg;           /* sched 7: users per scanned run of the served slice; sched 11: samples per batch (rs_carve) */
  int32_t sched;
  int32_t n_cells, n_ttis;
  int32_t refresh, phy_draws;
  int32_t epoch_wrap;        /* 1: the epoch grids cycle (epoch index modulo n_epochs) instead of ending the run */
  int32_t direct;            /* 1: rs_schedule_tti -- avg/rand given, no EWMA, no clock */
  int32_t rand0, rand1;      /* direct mode */
  /* configuration (device pointers) */
  const RsTables* tab;
  const double* weight;      /* [S] */
  const int32_t* eps;        /* [S] */
  const int32_t* psi;        /* [S] */
  const uint8_t* user_slice; /* [U] */
  const uint8_t* prb_cqi;    /* direct mode, optional: [U][R*G] per-PRB CQI for the EESM sum */
  /* direct mode, customised slices (alpha = 1): */
  int32_t queue_mode;        /* 1: some slice has alpha != 0 */
  const int32_t* alpha;      /* [S] */
  const int32_t* beta;       /* [S] */
  const double* hol;         /* [U] head-of-line delay of the slice-priority bearer */
  const uint8_t* prio;       /* [U] prioritized bearer has data (NULL = all) */
  /* finite queues (batched mode, rs_batch_set_bearers / rs_batch_set_arrivals): two bearers per user, index = priority */
  const uint8_t* bearer_kind; /* [U][2] 0 none, 1 InfiniteBuffer, 2 finite queue */
  const int64_t* arr_off;     /* [n_cells*U*2 + 1] first arrival burst of every bearer in the arrays below */
  const double* arr_time;     /* burst time stamps (ascending per bearer) */
  const int32_t* arr_nfull;   /* full packets (RS_FULL_PACKET bytes) of the burst */
  const int32_t* arr_last;    /* bytes of the burst's last packet (0: none) */
  int32_t *q_head, *q_tail, *q_pk, *q_frag, *q_bytes, *q_pkts, *b_tx; /* [cells][2][U] queue window / progress / totals, bytes since the last EWMA */
  double* b_avg;              /* [cells][2][U] RadioBearer::m_averageTransmissionRate */
  int64_t *b_cumb, *b_cumr;   /* [cells][2][U] m_cumulativeBytes / m_cumulativeRBs */
  uint8_t* q_flags;           /* [cells][U] bit 0: prioritized bearer has data, bit 1: user has queued data */
  double* q_hol;              /* [cells][U] head-of-line delay of the slice-priority bearer */
  int32_t exact_scan;        /* drop-in mode: an input lies outside the FP32 filter's safe range -> every user is compared exactly */
  int32_t synthetic;         /* rs_config.synthetic_exp: transport blocks PRB by PRB (schedulers 7, 8, 9, 10, 101, 103) */
  int32_t gen_exp;           /* drop-in mode: some slice has algo_epsilon / algo_psi outside {0, 1}: `avg` holds pow(avg_kbps, psi) as the
                              * host's libm gave it, gen_num the numerators, and every user is compared exactly */
  const double* gen_num;     /* [S][16] pow(eff(cqi) * 180000 / 1000, algo_epsilon of the slice), host libm */
  const int32_t* gate;       /* drop-in mode, optional [U]: m_requiredRBs (sched 7) / dataToTransmit bytes (sched 1); NULL = backlogged */
  int32_t* log_upper;        /* sched 10, drop-in mode: [S][R] (rbg | user << 8), -1 padded; NULL = off */
  const uint8_t* draws;      /* sched 11, drop-in mode: rand() % 4 of the RS_NVS_SAMPLES x U draws, in draw order */
  const int32_t* tbs_eff;    /* [R+1][16] TBS bits of n RBGs (n*G PRBs) at a final CQI (the CQI -> MCS -> I_TBS step folded in by the host), incl. the >110-PRB rule */
  /* state */
  double* avg;               /* [cells][U] */
  int32_t* tx_bytes;         /* [cells][U] */
  int64_t* cum_bytes;        /* [cells][U] */
  int64_t* cum_rbs;          /* [cells][U] */
  double* slice_state;       /* [cells][S] slice_rbs_offset_ | slice_ewma_time_ */
  RsCellScalars* scal;       /* [cells] */
  /* CQI sources */
  int32_t cqi_mode;
  const uint8_t* epochs;     /* [cells][n_epochs][grid_stride] */
  int64_t grid_stride;       /* bytes per grid, multiple of 16 */
  int32_t n_epochs;
  const uint8_t* trace;      /* [n_traces][n_rows][R] */
  int32_t n_traces, n_rows, row_mod;
  const int32_t* user_trace; /* [cells][U] */
  /* optional per-PRB twins of the two sources (link adaptation reads them; the metric reads the per-RBG arrays above) */
  const uint8_t* epochs_prb; /* [cells][n_epochs][grid_stride_prb]: [U][R*G] */
  int64_t grid_stride_prb;
  const uint8_t* trace_prb;  /* [n_traces][n_rows][R*G] */
  /* optional per-TTI log */
  int16_t* log_map;          /* [cells][n_ttis][R] */
  int16_t* log_quota;        /* [cells][n_ttis][S] */
  int16_t* log_target;       /* [cells][n_ttis][S] */
  int32_t* log_tbs;          /* [cells][n_ttis][U], pre-zeroed */
  int32_t* log_uinfo;        /* [cells][n_ttis][U], pre-zeroed: nprb | final_cqi<<16 | mcs<<24 */
  uint32_t* log_keys;        /* [cells][n_ttis][R][S] transport schedulers: CQI key of the slice's best user | (user+1)<<8 */
  int32_t* err;              /* device error word */
  unsigned long long* stamps; /* diagnostic build (-DRS_STAMPS): [cells][20] phase cycles, else unused */
  /* LDS carve (byte offsets from the dynamic LDS base) */
  int32_t off_avgk, off_rcp, off_tx, off_tab, off_slice, off_items, off_elems,
      off_sorted, off_sortx, off_misc, off_tbs, off_cqi, off_queue, off_qstate, q_lds, lds_bytes;
  int32_t n_seg, n_items;    /* segments per RBG scan, R*n_seg */
  /* drop-in mode, optional: a word in the caller's pinned output block that the kernel's very last store sets to done_seq (system
   * scope, after every thread's outputs): rs_schedule_tti polls it instead of waiting for the stream's completion signal */
  uint32_t* done_flag;
  uint32_t done_seq;
  /* drop-in mode, rs_tti_in.cqi_epoch (round 6): the context keeps the call's CQI grid on the device in the layout of the LDS grid
   * (RBG-major [R][Upad], zero padding included).  image_mode 0: no image (cqi_epoch = 0); 1: the reports changed -- transpose the
   * caller's [U][R] block as always, then store the LDS grid to grid_image; 2: same reports as the call before -- a straight 16-byte
   * copy of grid_image into LDS, the caller's block is not read (39 calls of 40 in the reference: CQI_INTERVAL 40) */
  uint8_t* grid_image;
  int32_t image_mode;
  /* batches: cells per dispatch round = the device's compute units, when the batch puts exactly two cells on every CU (0: no
   * priority balancing between co-resident cells, RS_SETPRIO in rs_kernels.hip) */
  int32_t prio_round_cells;
  /* ... or, preferred: feedback.  *prio_sum = TTIs done by all cells of this launch together (every cell adds RS_PRIO_PERIOD each
   * RS_PRIO_PERIOD TTIs); a cell whose own count times n_cells is below it is behind the average and runs boosted until the next
   * look.  nullptr: off.  Zeroed by the host before every launch. */
  unsigned long long* prio_sum;
};

#endif /* RS_DEVICE_H_ */
