// Drop-in mode end to end through the C++ adapter (include/radiosaber_scheduler.hpp), driven the
// way the reference drives its scheduler: simulator clock by repeated addition, CQI reports every
// 40 TTIs from the trace rows, the REAL libc rand() stream shared with the PHY error model.
// argv: trace.bin (u8 [100 users][n_rows][64], already mapped per user)  n_rows  rand_skip  n_ttis
// Prints "cumu <user> <bytes> <rbs>" for users 1, 2, 5 and the first TTI's allocations.
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/radiosaber_scheduler.hpp"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  const int U = 100, R = 64, n_rows = atoi(argv[2]);
  const long skip = atol(argv[3]);
  const int n_ttis = atoi(argv[4]);
  const int sched_id = argc > 5 ? atoi(argv[5]) : RS_SCHED_MAXCELL;
  std::vector<uint8_t> trace((size_t)U * n_rows * R);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(trace.data(), 1, trace.size(), f) != trace.size()) return 3;
  fclose(f);
  std::vector<int> ues(20, 5), zeros(20, 0), ones(20, 1);
  std::vector<double> w(20, 0.05);
  radiosaber::GpuDownlinkScheduler sched(ues, w, zeros, zeros, ones, ones, 512, 8, sched_id);
  std::ostringstream log_out, log_err;
  sched.SetTimeStamp(100);  // the reference's m_ts has counted the 100 idle TTIs before the first allocation
  sched.SetLogStreams(&log_out, &log_err);
  srand(805290992);  // seed.h commonSeed[0]
  for (long i = 0; i < skip; i++) (void)rand();
  double t = 0;
  for (int k = 0; k < 100; k++) t += 0.001;
  for (int u = 0; u < U; u++) sched.Bearer(u).last_update = 0.1;  // bearers created at application start
  long last_sent = 0;
  bool reported = false;
  size_t served_prev = 0;
  for (int n = 0; n < n_ttis; n++) {
    for (size_t i = 0; i < served_prev; i++) (void)rand();  // wideband-cqi-eesm-error-model.cpp:69
    if (!reported || ((int)(t * 1000) - last_sent) >= 40) {
      reported = true;
      last_sent = (long)(t * 1000);
      int row = ((int)(t * 1000 / 40)) % 475;
      if (row >= n_rows) return 4;
      for (int u = 0; u < U; u++) sched.SetCQI(u, &trace[((size_t)u * n_rows + row) * R]);
    }
    sched.DoSchedule(t);
    served_prev = sched.LastAllocations().size();
    if (n == 0) {
      for (const auto& a : sched.LastAllocations())
        if (a.user_id == 1 || a.user_id == 4 || a.user_id == 47) {
          printf("first %d final_cqi %d rbgs", a.user_id, a.final_cqi);
          for (size_t i = 0; i < a.prbs.size(); i += 8) printf(" %d", a.prbs[i] / 8);
          printf("\n");
        }
      printf("served %zu quota16 %d %d\n", served_prev, sched.SliceTargetRbs()[16], sched.SliceQuotaRbgs()[16]);
    }
    t += 0.001;
  }
  for (int u : {1, 2, 5}) printf("cumu %d %lu %lu\n", u, sched.Bearer(u).cumulative_bytes, sched.Bearer(u).cumulative_rbs);
  printf("ts %lu\n", sched.GetTimeStamp());
  for (int u = 0; u < U; u++) printf("ALL %d %lu %lu\n", u, sched.Bearer(u).cumulative_bytes, sched.Bearer(u).cumulative_rbs);
  // the reference-format logs, for the golden lines of SURVEY.md Appendix A
  {
    std::istringstream is(log_out.str());
    std::string line;
    int n = 0;
    while (std::getline(is, line) && n < 60) { printf("OUT %s\n", line.c_str()); n++; }
    std::istringstream es(log_err.str());
    while (std::getline(es, line))
      if (line.rfind("100 ", 0) == 0 || line.rfind("299 ", 0) == 0) printf("ERR %s\n", line.c_str());
  }
  return 0;
}
