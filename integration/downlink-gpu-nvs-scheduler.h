/* downlink-gpu-nvs-scheduler.h -- DownlinkNVSScheduler (CLI schedulers 7 and 11) with the RBG allocation on an MI355X.
 * Copy next to downlink-nvs-scheduler.h.  Needs the reference tree's generated load-parameters.h and jsoncpp to compile.
 * SelectSliceToServe (O(S), carries slice_ewma_time_), the EWMA, SelectFlowsToSchedule and DoStopSchedule stay the parent's.
 * RBsAllocation() is virtual and simply overridden; RBsAllocationNonGreedyPF() is not, so DoSchedule() (virtual) is
 * overridden too and repeats the parent's sequence (downlink-nvs-scheduler.cpp:196-218) with the GPU call in its place. */
#ifndef DOWNLINKGPUNVSSCHEDULER_H_
#define DOWNLINKGPUNVSSCHEDULER_H_

#include <string>
#include <vector>

#include "downlink-nvs-scheduler.h"
#include "radiosaber_hip.h"

class DownlinkGpuNVSScheduler : public DownlinkNVSScheduler {
 public:
  DownlinkGpuNVSScheduler(std::string config_fname, bool is_nongreedy, int hip_device = 0);
  virtual ~DownlinkGpuNVSScheduler();

  virtual void DoSchedule(void);
  virtual void RBsAllocation(); /* greedy (7) or sampled (11), by the constructor's flag */

 private:
  rs_ctx* ctx_;
  int hip_device_;
  bool nongreedy_; /* the parent's is_nongreedy_ is private */
  int num_slices_, nb_rbs_;
  std::vector<int> user_to_slice_;
  std::vector<double> slice_weights_;
  std::vector<int> alpha_, beta_, epsilon_, psi_;
  std::vector<uint8_t> cqi_prb_;    /* the previous call's per-PRB reports (rs_tti_in.cqi_epoch) */
  unsigned long long cqi_epoch_;    /* bumped when any report changed */
};

#endif /* DOWNLINKGPUNVSSCHEDULER_H_ */
