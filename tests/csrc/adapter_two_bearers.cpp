// Customised slices with two bearers per user through the C++ adapter (include/radiosaber_scheduler.hpp):
// SelectFlowsToSchedule / InsertFlowToUser (bearer index = priority, slice priority = highest priority with packets),
// the metric's "1 + sum of the bearers' averages", the head-of-line delay and "prioritized bearer has data" inputs, and
// DoStopSchedule's split of the granted bytes from the highest priority down.  The scenario is a fixed script that
// tests/test_adapter.py replays with the oracle and plain Python bookkeeping.
// argv: sched n_ttis.  Prints per TTI "T <n> <user>:<nprb>:<final_cqi>:<tbs> ...", then per bearer
// "B <user> <prio> <cumu_bytes> <cumu_rbs> <avg as %a>", then the stderr-format lines as "ERR ...".
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/radiosaber_scheduler.hpp"

int main(int argc, char** argv) {
  const int sched_id = argc > 1 ? atoi(argv[1]) : RS_SCHED_MAXCELL;
  const int n_ttis = argc > 2 ? atoi(argv[2]) : 60;
  const std::vector<int> ues = {3, 4, 3, 2}, alpha = {0, 1, 1, 0}, beta = {0, 0, 1, 0}, eps = {1, 1, 1, 1}, psi = {1, 1, 1, 0};
  const std::vector<double> w = {0.3, 0.3, 0.2, 0.2};
  const int U = 12, R = 25, G = 4;
  radiosaber::GpuDownlinkScheduler sched(ues, w, alpha, beta, eps, psi, R * G, G, sched_id);
  std::ostringstream log_err;
  sched.SetLogStreams(nullptr, &log_err);
  // second bearers: every user of the customised slices 1 and 2 (users 3..9) and user 0 of the plain slice 0
  std::vector<int> second = {0, 3, 4, 5, 6, 7, 8, 9};
  for (int u : second) sched.AddBearer(u, 1, 100 + u).last_update = 0.1;
  for (int u = 0; u < U; u++) sched.Bearer(u).last_update = 0.1;
  srand(1234);
  double t = 0.1;
  std::vector<uint8_t> cqi(R);
  for (int n = 0; n < n_ttis; n++) {
    if (n % 10 == 0)
      for (int u = 0; u < U; u++) {
        for (int r = 0; r < R; r++) cqi[r] = (uint8_t)(1 + ((unsigned)(u * 131 + r * 37 + (n / 10) * 101) * 2654435761u >> 7) % 15);
        sched.SetCQI(u, cqi.data());
      }
    for (int u : second) {
      radiosaber::BearerState& b = sched.Bearer(u, 1);
      b.queue_size = ((u * 7 + n * 13) % 5 == 0) ? 0 : 200 + ((u * 31 + n * 17) % 1500);
      b.has_packets = b.queue_size > 0;
      b.hol_delay = 0.001 * (1 + (u * 5 + n * 3) % 40);
    }
    // user 4's best-effort bearer is idle every third TTI: only its priority-1 bearer is in the record then (if it has data)
    sched.Bearer(4, 0).has_packets = (n % 3 != 0) || !sched.Bearer(4, 1).has_packets;
    sched.DoSchedule(t);
    printf("T %d", n);
    for (const auto& a : sched.LastAllocations()) printf(" %d:%d:%d:%d", a.user_id, a.n_prbs, a.final_cqi, a.tbs_bits);
    printf("\n");
    t += 0.001;
  }
  for (int u = 0; u < U; u++)
    for (int pr = 0; pr < 2; pr++) {
      const radiosaber::BearerState& b = sched.Bearer(u, pr);
      if (b.exists) printf("B %d %d %lu %lu %a\n", u, pr, b.cumulative_bytes, b.cumulative_rbs, b.average_transmission_rate);
    }
  std::istringstream es(log_err.str());
  std::string line;
  while (std::getline(es, line)) printf("ERR %s\n", line.c_str());
  return 0;
}
