/* Latency of the drop-in entry point from C++ (no Python in the loop): rs_schedule_tti per call, host buffers in and out.
 * Build: g++ -O2 -std=c++17 -Iinclude tools/dropin_latency.cpp -Lradiosaber_amd -lradiosaber_hip -Wl,-rpath,'$ORIGIN/../radiosaber_amd' -o tools/dropin_latency
 * Run on a GPU box: tools/dropin_latency [calls] */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "radiosaber_hip.h"

static double run(int ues_per_slice, int R, int G, int sched, int calls, bool specialise, bool epoch = false) {
  const int S = 20, U = S * ues_per_slice;
  std::vector<double> w(S, 0.05);
  std::vector<int32_t> zero(S, 0), one(S, 1), u2s(U);
  for (int i = 0; i < U; i++) u2s[i] = i / ues_per_slice;
  rs_config cfg{};
  cfg.n_slices = S;
  cfg.n_users = U;
  cfg.n_rbgs = R;
  cfg.rbg_size = G;
  cfg.sched = sched;
  cfg.device = 0;
  cfg.slice_weight = w.data();
  cfg.algo_alpha = zero.data();
  cfg.algo_beta = zero.data();
  cfg.algo_epsilon = one.data();
  cfg.algo_psi = one.data();
  cfg.user_to_slice = u2s.data();
  rs_ctx* c = RS_CREATE(&cfg);
  if (!c) { fprintf(stderr, "rs_create: %s\n", rs_last_error()); exit(1); }
  if (specialise && rs_ctx_specialize(c) != RS_OK) { fprintf(stderr, "rs_ctx_specialize: %s\n", rs_last_error()); exit(1); }
  std::mt19937 g(1);
  std::vector<uint8_t> cqi((size_t)U * R);
  std::vector<double> avg(U);
  for (auto& x : cqi) x = 1 + g() % 15;
  std::vector<uint8_t> grids(4 * cqi.size());
  for (auto& x : grids) x = 1 + g() % 15;
  for (auto& x : avg) x = 1e4 + g() % 1000000;
  std::vector<int32_t> map(R), tbs(U), nprb(U), fcqi(U), mcs(U), tgt(S), quo(S);
  rs_tti_in in{};
  in.n_users = sched == RS_SCHED_NVS ? ues_per_slice : U; /* NVS: the caller passes the served slice's users (here slice 0) */
  in.cqi = cqi.data();
  in.avg_rate = avg.data();
  rs_tti_out out{};
  out.rbg_to_user = map.data();
  out.user_tbs_bits = tbs.data();
  out.user_nprb = nprb.data();
  out.user_final_cqi = fcqi.data();
  out.user_mcs = mcs.data();
  out.target_rbs = tgt.data();
  out.quota_rbgs = quo.data();
  long long sum = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = -50; i < calls; i++) {
    if (i == 0) t0 = std::chrono::steady_clock::now();
    in.rand0 = 123 + i;
    in.rand1 = 456 + i;
    avg[(i + 50) % U] += 1000; /* inputs change call to call */
    /* rs_tti_in.cqi_epoch: the reports change every 40 calls (CQI_INTERVAL, enb-mac-entity.cc:38); in between the context's device image serves */
    if (epoch && (i + 50) % 40 == 0) memcpy(cqi.data(), grids.data() + (size_t)(((i + 50) / 40) % 4) * cqi.size(), cqi.size()); /* (pre-drawn: drawing 12.5 K values costs the host ~60 us) */
    in.cqi_epoch = epoch ? 1 + (uint64_t)(i + 50) / 40 : 0;
    if (rs_schedule_tti(c, &in, &out) != RS_OK) { fprintf(stderr, "rs_schedule_tti: %s\n", rs_last_error()); exit(1); }
    sum += map[0] + tbs[map[0] < 0 ? 0 : map[0]];
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / calls;
  rs_destroy(c);
  printf("sched %d, %d UEs x %d RBGs, %s kernel%s: %.1f us per rs_schedule_tti (checksum %lld)\n", sched, U, R,
         specialise ? "shape-specialised" : "built-in", epoch ? ", cqi_epoch (new reports every 40 calls)" : "", us, sum);
  return us;
}

int main(int argc, char** argv) {
  const int calls = argc > 1 ? atoi(argv[1]) : 2000;
  for (int sp = 0; sp < 2; ++sp) { /* built-in kernels, then the context's own build (rs_ctx_specialize) */
    run(5, 64, 8, RS_SCHED_MAXCELL, calls, sp != 0);
    run(25, 25, 4, RS_SCHED_MAXCELL, calls, sp != 0);
    run(25, 64, 8, RS_SCHED_MAXCELL, calls, sp != 0);
    run(25, 25, 4, RS_SCHED_SEQUENTIAL, calls, sp != 0);
    run(25, 25, 4, RS_SCHED_PF, calls, sp != 0);
    run(25, 25, 4, RS_SCHED_NVS, calls, sp != 0);
  }
  /* round 6: the caller says when its reports changed (rs_tti_in.cqi_epoch) */
  for (int sp = 0; sp < 2; ++sp) {
    run(5, 64, 8, RS_SCHED_MAXCELL, calls, sp != 0, true);
    run(25, 25, 4, RS_SCHED_MAXCELL, calls, sp != 0, true);
    run(25, 64, 8, RS_SCHED_MAXCELL, calls, sp != 0, true);
    run(25, 25, 4, RS_SCHED_SEQUENTIAL, calls, sp != 0, true);
    run(25, 25, 4, RS_SCHED_PF, calls, sp != 0, true);
  }
  return 0;
}
