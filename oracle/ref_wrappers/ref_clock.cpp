// Wrapper translation unit: compiles the reference's discrete-event core WHERE IT LIES
// (/root/reference/src/core/eventScheduler/{simulator.cc,calendar.cpp,event.cpp}, added with -I by
// oracle/Makefile) and drives it the way the simulator's frame loop does, to pin the simulated clock the PF
// EWMA divides by (t_k = fl(t_{k-1} + 0.001), simulator.cc:117-126) and the ordering of events with equal
// time stamps (calendar.cpp:58-68).  No reference source is copied here.
//
// The ticker below restates only the *scheduling pattern* of componentManagers/FrameManager.cpp:118-189
// (StartSubframe schedules StopSubframe at +0.001, StopSubframe schedules the next StartSubframe at +0.0) and of
// flows/application/Application.cpp:321-327 (one event at an absolute offset from t = 0); every time stamp it
// records comes out of the reference's own Simulator::DoSchedule / Calendar.
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "core/eventScheduler/simulator.cc"
#include "core/eventScheduler/calendar.cpp"
#include "core/eventScheduler/event.cpp"

namespace {
struct Ticker {
  double* out;
  int n, k;
  double app_start_seen;
  int app_order; /* number of subframe starts that ran before the application-start event */
  void StartSubframe() {
    if (k < n) out[k] = Simulator::Init()->Now();
    ++k;
    if (k < n) Simulator::Init()->Schedule(0.001, &Ticker::StopSubframe, this);
  }
  void StopSubframe() { Simulator::Init()->Schedule(0.0, &Ticker::StartSubframe, this); }
  void AppStart() {
    app_start_seen = Simulator::Init()->Now();
    app_order = k;
  }
};
}  // namespace

// Runs n subframes from t = 0 with one "application start" event scheduled at `app_start` seconds (from t = 0, before
// the frame loop starts, like SingleCellWithInterference does).  out[k] = Simulator::Now() at the start of subframe k;
// *app_now = Now() inside the application-start event; *app_before = subframe starts that ran before it.
// The reference's Simulator is a process-wide singleton whose clock cannot be reset, so every call runs in a forked
// child (the parent never instantiates it) and hands the time stamps back through shared memory.
static int run_in_this_process(int n, double app_start, double* out, double* app_now, int* app_before) {
  Ticker t{out, n, 0, -1.0, -1};
  Simulator* sim = Simulator::Init();
  sim->Schedule(app_start, &Ticker::AppStart, &t);
  sim->Schedule(0.0, &Ticker::StartSubframe, &t);
  sim->Run();
  *app_now = t.app_start_seen;
  *app_before = t.app_order;
  return t.k;
}

extern "C" int ref_clock_run(int n, double app_start, double* out, double* app_now, int* app_before) {
  if (n < 1) return -1;
  const size_t bytes = sizeof(double) * ((size_t)n + 1) + sizeof(int) * 2;
  void* shm = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (shm == MAP_FAILED) return -2;
  double* s_out = (double*)shm;
  int* s_int = (int*)(s_out + n + 1);
  const pid_t pid = fork();
  if (pid < 0) { munmap(shm, bytes); return -3; }
  if (pid == 0) {
    s_int[1] = run_in_this_process(n, app_start, s_out, s_out + n, s_int);
    _exit(0);
  }
  int status = 0;
  waitpid(pid, &status, 0);
  int rc = -4;
  if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
    for (int k = 0; k < n; k++) out[k] = s_out[k];
    *app_now = s_out[n];
    *app_before = s_int[0];
    rc = s_int[1];
  }
  munmap(shm, bytes);
  return rc;
}
