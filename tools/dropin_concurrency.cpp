/* The drop-in entry point under the reference's own launch pattern: K simulators at once.
 *
 * The reference scales by starting one OS process per (scheduler, seed, mapping) with `&`
 * (NSDI23-radiosaber-experiments/exp-customization/run_backlogged.sh:6-14, exp-fix20slices/run_exps.sh: 3 x 9 = 27 at a time); each of
 * them calls RBsAllocation() once per TTI (downlink-transport-scheduler.cpp:453-675).  With the GPU scheduler linked in, that is K
 * rs_ctx on one MI355X, each calling rs_schedule_tti back to back.  This program measures exactly that, two ways:
 *
 *   dropin_concurrency threads K SHAPE [calls] [epoch|noepoch] [builtin]     K contexts on K host threads of ONE process
 *   dropin_concurrency procs   K SHAPE [calls] [epoch|noepoch] [builtin]     K processes, one context each (fork + exec of this binary
 *                                                                            from a parent that never touches the GPU)
 *   SHAPE: 500x25 (20 slices x 25 UEs, 25 RBGs of 4 PRBs) | 100x64 (20 x 5 UEs, 64 RBGs of 8: the shipped exp-fix20slices/5ues shape)
 *   further options: think=US (host time between two calls of a worker: the simulator's own work per TTI; default 0 = back to back),
 *                    hwq=N (GPU_MAX_HW_QUEUES=N before the first HIP call: ROCclr maps a process's streams onto 4 hardware queues by default),
 *                    sched=N
 *
 * Every worker: rs_create, rs_ctx_specialize (unless `builtin`), 60 warm-up calls (they include the specialised build's checked calls,
 * see rs_ctx_jit_status), then -- all workers released together -- `calls` timed calls (default 2 000).  The CQI block changes every 40
 * calls (CQI_INTERVAL, enb-mac-entity.cc:38) and `epoch` (default) says so through rs_tti_in.cqi_epoch; the PF averages change on every
 * call.  Output: one line per run with the pooled per-call p50 / p90 / p99 / max in microseconds and the aggregate TTIs/s
 * (K x calls / (last end - first start)).
 *
 * Build: g++ -O2 -std=c++17 -pthread -Iinclude tools/dropin_concurrency.cpp -Lradiosaber_amd -lradiosaber_hip \
 *            -Wl,-rpath,'$ORIGIN/../radiosaber_amd' -o tools/dropin_concurrency
 */
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "radiosaber_hip.h"

namespace {
using clk = std::chrono::steady_clock;

struct Shape { int ues_per_slice, R, G; };
bool parse_shape(const char* s, Shape* out) {
  if (!strcmp(s, "500x25")) { *out = {25, 25, 4}; return true; }
  if (!strcmp(s, "100x64")) { *out = {5, 64, 8}; return true; }
  if (!strcmp(s, "500x64")) { *out = {25, 64, 8}; return true; }
  return false;
}

struct Worker {
  Shape sh;
  int calls = 2000, sched = RS_SCHED_MAXCELL, id = 0;
  int think_us = 0; /* host time between two calls (the simulator's own work per TTI): 0 = back to back */
  bool epoch = true, specialise = true;
  std::vector<float> us;      /* per-call latency */
  double t_first = 0, t_last = 0; /* seconds on CLOCK_MONOTONIC (the same clock in every process of the machine) */
  long long checksum = 0;
  std::string error, status;

  rs_ctx* c = nullptr;
  std::vector<double> w, avg;
  std::vector<int32_t> zero, one, u2s, map, tbs, nprb, fcqi, mcs, tgt, quo;
  std::vector<uint8_t> cqi, grids; /* four pre-drawn report sets: drawing one costs the host ~60 us, which is the simulator's time, not the call's */
  std::mt19937 g;
  rs_tti_in in{};
  rs_tti_out out{};
  uint64_t ep = 0;
  int n_done = 0;

  bool setup() {
    const int S = 20, U = S * sh.ues_per_slice;
    w.assign(S, 0.05); zero.assign(S, 0); one.assign(S, 1); u2s.resize(U);
    for (int i = 0; i < U; i++) u2s[i] = i / sh.ues_per_slice;
    rs_config cfg{};
    cfg.n_slices = S; cfg.n_users = U; cfg.n_rbgs = sh.R; cfg.rbg_size = sh.G; cfg.sched = sched; cfg.device = 0;
    cfg.slice_weight = w.data(); cfg.algo_alpha = zero.data(); cfg.algo_beta = zero.data();
    cfg.algo_epsilon = one.data(); cfg.algo_psi = one.data(); cfg.user_to_slice = u2s.data();
    c = RS_CREATE(&cfg);
    if (!c) { error = std::string("rs_create: ") + rs_last_error(); return false; }
    if (specialise && rs_ctx_specialize(c) != RS_OK) { error = std::string("rs_ctx_specialize: ") + rs_last_error(); return false; }
    g.seed(1000 + id);
    cqi.resize((size_t)U * sh.R); avg.resize(U);
    grids.resize(4 * cqi.size());
    for (auto& x : grids) x = 1 + g() % 15;
    for (auto& x : avg) x = 1e4 + g() % 1000000;
    map.resize(sh.R); tbs.resize(U); nprb.resize(U); fcqi.resize(U); mcs.resize(U); tgt.resize(S); quo.resize(S);
    in.n_users = U; in.cqi = cqi.data(); in.avg_rate = avg.data();
    out.rbg_to_user = map.data(); out.user_tbs_bits = tbs.data(); out.user_nprb = nprb.data(); out.user_final_cqi = fcqi.data();
    out.user_mcs = mcs.data(); out.target_rbs = tgt.data(); out.quota_rbgs = quo.data();
    return true;
  }
  bool one_call() {
    if (n_done % 40 == 0) { /* new reports: the caller's block changes, as a UE's GetCqiFeedbacks() vector does every CQI_INTERVAL */
      memcpy(cqi.data(), grids.data() + (size_t)((n_done / 40) % 4) * cqi.size(), cqi.size());
      ++ep;
    }
    in.cqi_epoch = epoch ? ep : 0;
    in.rand0 = (int)(g() >> 1); in.rand1 = (int)(g() >> 1);
    avg[n_done % avg.size()] += 1000;
    if (rs_schedule_tti(c, &in, &out) != RS_OK) { error = std::string("rs_schedule_tti: ") + rs_last_error(); return false; }
    checksum += map[0] + tbs[map[0] < 0 ? 0 : map[0]];
    ++n_done;
    return true;
  }
  bool warm() {
    for (int i = 0; i < 60; i++) if (!one_call()) return false;
    char msg[512] = "";
    const int code = rs_ctx_jit_status(c, msg, sizeof msg);
    status = std::to_string(code) + " " + msg;
    if (specialise && code != 1) { error = "specialised kernel not in use: " + status; return false; }
    return true;
  }
  bool timed() {
    us.resize(calls);
    const auto sec = [](clk::time_point t) { return std::chrono::duration<double>(t.time_since_epoch()).count(); };
    clk::time_point t0 = clk::now();
    t_first = sec(t0);
    for (int i = 0; i < calls; i++) {
      if (think_us > 0) { /* the simulator computes: the latency clock starts when it comes back to the scheduler */
        const clk::time_point until = t0 + std::chrono::microseconds(think_us);
        while (clk::now() < until) __builtin_ia32_pause();
        t0 = clk::now();
      }
      if (!one_call()) return false;
      const clk::time_point t1 = clk::now();
      us[i] = std::chrono::duration<float, std::micro>(t1 - t0).count();
      t0 = t1;
    }
    t_last = sec(t0);
    return true;
  }
  void close() { if (c) rs_destroy(c); c = nullptr; }
};

void report(const char* mode, int K, const char* shape, const Worker& proto, std::vector<float>& all, double first, double last, long long checksum,
            const std::string& status) {
  std::sort(all.begin(), all.end());
  auto pct = [&](double p) { return all.empty() ? 0.f : all[std::min(all.size() - 1, (size_t)(p * all.size()))]; };
  double mean = 0;
  for (float x : all) mean += x;
  mean /= all.empty() ? 1 : all.size();
  char extra[96] = "";
  if (proto.think_us) snprintf(extra, sizeof extra, " think=%dus", proto.think_us);
  if (const char* q = getenv("GPU_MAX_HW_QUEUES")) snprintf(extra + strlen(extra), sizeof extra - strlen(extra), " GPU_MAX_HW_QUEUES=%s", q);
  printf("%s K=%d %s %s %s%s calls=%d: per call p50 %.1f us, p90 %.1f, p99 %.1f, max %.1f, mean %.1f; aggregate %.0f TTIs/s over %.1f ms (checksum %lld; worker 0: %s)\n",
         mode, K, shape, proto.specialise ? "specialised" : "built-in", proto.epoch ? "cqi_epoch" : "no-epoch", extra, proto.calls, pct(0.5), pct(0.9), pct(0.99),
         all.empty() ? 0.f : all.back(), mean, all.size() / (last - first), (last - first) * 1e3, checksum, status.c_str());
  fflush(stdout);
}

int run_threads(int K, const char* shape, const Worker& proto) {
  std::vector<Worker> ws(K, proto);
  std::atomic<int> ready{0}, failed{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> th;
  for (int i = 0; i < K; i++) {
    ws[i].id = i;
    th.emplace_back([&, i] {
      Worker& w = ws[i];
      const bool ok = w.setup() && w.warm();
      if (!ok) failed++;
      ready++;
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      if (ok && !failed.load() && !w.timed()) failed++;
      w.close();
    });
  }
  while (ready.load() < K) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  go.store(true, std::memory_order_release);
  for (auto& t : th) t.join();
  if (failed.load()) {
    for (auto& w : ws) if (!w.error.empty()) fprintf(stderr, "worker %d: %s\n", w.id, w.error.c_str());
    return 1;
  }
  std::vector<float> all;
  double first = 1e300, last = 0;
  long long sum = 0;
  for (auto& w : ws) { all.insert(all.end(), w.us.begin(), w.us.end()); first = std::min(first, w.t_first); last = std::max(last, w.t_last); sum += w.checksum; }
  report("threads", K, shape, proto, all, first, last, sum, ws[0].status);
  return 0;
}

/* one process of `procs`: "ready" on stdout when warm, waits for a byte on stdin, runs, prints its numbers */
int run_worker(int id, const char* /*shape*/, Worker w) {
  w.id = id;
  if (!w.setup() || !w.warm()) { printf("error %s\n", w.error.c_str()); fflush(stdout); return 1; }
  printf("ready\n");
  fflush(stdout);
  char ch;
  if (read(0, &ch, 1) != 1 || ch != 'g') return 1;
  if (!w.timed()) { printf("error %s\n", w.error.c_str()); fflush(stdout); return 1; }
  printf("span %.9f %.9f %lld\n", w.t_first, w.t_last, w.checksum);
  printf("status %s\n", w.status.c_str());
  for (float x : w.us) printf("%.2f\n", x);
  printf("end\n");
  fflush(stdout);
  w.close();
  return 0;
}

int run_procs(int K, const char* shape, const Worker& proto, char** argv_tail, int n_tail) {
  /* the parent makes no HIP call: its children exec this binary afresh */
  struct Child { pid_t pid; int to, from; FILE* f; };
  std::vector<Child> cs;
  for (int i = 0; i < K; i++) {
    int to[2], from[2];
    if (pipe(to) || pipe(from)) { perror("pipe"); return 1; }
    const pid_t pid = fork();
    if (pid < 0) { perror("fork"); return 1; }
    if (pid == 0) {
      dup2(to[0], 0); dup2(from[1], 1);
      close(to[0]); close(to[1]); close(from[0]); close(from[1]);
      for (auto& c : cs) { close(c.to); close(c.from); }
      std::vector<char*> av;
      std::string idarg = std::to_string(i);
      av.push_back((char*)"dropin_concurrency"); av.push_back((char*)"worker"); av.push_back((char*)idarg.c_str());
      for (int k = 0; k < n_tail; k++) av.push_back(argv_tail[k]);
      av.push_back(nullptr);
      execv("/proc/self/exe", av.data());
      perror("execv");
      _exit(127);
    }
    close(to[0]); close(from[1]);
    cs.push_back({pid, to[1], from[0], fdopen(from[0], "r")});
  }
  char line[1024];
  bool bad = false;
  for (auto& c : cs) { /* every worker warm? */
    if (!fgets(line, sizeof line, c.f) || strncmp(line, "ready", 5) != 0) { fprintf(stderr, "worker %d: %s", (int)c.pid, line); bad = true; }
  }
  for (auto& c : cs) { const char go = bad ? 'x' : 'g'; if (write(c.to, &go, 1) != 1) bad = true; if (bad) close(c.to); }
  std::vector<float> all;
  double first = 1e300, last = 0;
  long long sum = 0;
  std::string status0;
  for (size_t i = 0; i < cs.size() && !bad; i++) {
    while (fgets(line, sizeof line, cs[i].f)) {
      double a, b; long long s;
      if (!strncmp(line, "end", 3)) break;
      if (!strncmp(line, "error", 5)) { fprintf(stderr, "worker %zu: %s", i, line); bad = true; break; }
      if (sscanf(line, "span %lf %lf %lld", &a, &b, &s) == 3) { first = std::min(first, a); last = std::max(last, b); sum += s; continue; }
      if (!strncmp(line, "status ", 7)) { if (i == 0) { status0 = line + 7; if (!status0.empty() && status0.back() == '\n') status0.pop_back(); } continue; }
      all.push_back((float)atof(line));
    }
  }
  for (auto& c : cs) { int st; waitpid(c.pid, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st)) bad = true; fclose(c.f); }
  if (bad) return 1;
  report("procs", K, shape, proto, all, first, last, sum, status0);
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s threads|procs K 500x25|100x64|500x64 [calls] [epoch|noepoch] [builtin] [sched=N]\n", argv[0]); return 2; }
  const std::string mode = argv[1];
  const int K = atoi(argv[2]);
  Worker proto;
  if (!parse_shape(argv[3], &proto.sh)) { fprintf(stderr, "unknown shape %s\n", argv[3]); return 2; }
  for (int i = 4; i < argc; i++) {
    if (!strcmp(argv[i], "epoch")) proto.epoch = true;
    else if (!strcmp(argv[i], "noepoch")) proto.epoch = false;
    else if (!strcmp(argv[i], "builtin")) proto.specialise = false;
    else if (!strncmp(argv[i], "sched=", 6)) proto.sched = atoi(argv[i] + 6);
    else if (!strncmp(argv[i], "think=", 6)) proto.think_us = atoi(argv[i] + 6);
    /* ROCclr multiplexes a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: set before the first HIP call */
    else if (!strncmp(argv[i], "hwq=", 4)) setenv("GPU_MAX_HW_QUEUES", argv[i] + 4, 1);
    else if (atoi(argv[i]) > 0) proto.calls = atoi(argv[i]);
  }
  if (mode == "worker") return run_worker(K /* = id */, argv[3], proto);
  if (K < 1 || K > 256) { fprintf(stderr, "K = %d?\n", K); return 2; }
  if (mode == "threads") return run_threads(K, argv[3], proto);
  if (mode == "procs") return run_procs(K, argv[3], proto, argv + 3, argc - 3);
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
