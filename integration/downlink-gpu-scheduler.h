/* downlink-gpu-scheduler.h -- DownlinkTransportScheduler with RBsAllocation() on an MI355X (libradiosaber_hip.so).
 *
 * Copy next to downlink-transport-scheduler.h (src/protocolStack/mac/packet-scheduler/).  Needs the reference tree's
 * generated src/load-parameters.h and jsoncpp to compile, like every scheduler of the reference.
 *
 * Only RBsAllocation() is overridden: DoSchedule, UpdateAverageTransmissionRate, SelectFlowsToSchedule and
 * DoStopSchedule stay the parent's (downlink-transport-scheduler.cpp:105-221), so bearers, queues, RLC and the log
 * lines are untouched.  The parent's slice configuration is private (downlink-transport-scheduler.h:31-41), so this class
 * parses the same JSON keys again (:55-97); slice_rbs_offset_ lives in the GPU context. */
#ifndef DOWNLINKGPUSCHEDULER_H_
#define DOWNLINKGPUSCHEDULER_H_

#include <string>
#include <vector>

#include "downlink-transport-scheduler.h"
#include "radiosaber_hip.h"

class DownlinkGpuScheduler : public DownlinkTransportScheduler {
 public:
  /* interslice_algo as the parent takes it: 0 GreedyByRow (CLI 8), 1 SubOpt, 2 MaximizeCell (CLI 9), 3 Vogel, 4 UpperBound (CLI 10) */
  DownlinkGpuScheduler(std::string config_fname, int interslice_algo, int hip_device = 0);
  virtual ~DownlinkGpuScheduler();

  virtual void RBsAllocation();

 private:
  void LazyCreate(int nb_rbs, int rbg_size);

  rs_ctx* ctx_;
  int hip_device_;
  int sched_;       /* RS_SCHED_* */
  int num_slices_;  /* the parent's num_slices_ is private */
  int nb_rbs_, rbg_size_;
  bool any_alpha_;
  std::vector<int> user_to_slice_;
  std::vector<double> slice_weights_;
  std::vector<int> alpha_, beta_, epsilon_, psi_;
  std::vector<uint8_t> cqi_prb_;    /* the previous TTI's per-PRB reports (rs_tti_in.cqi_epoch) */
  unsigned long long cqi_epoch_;    /* bumped when any report changed */
};

#endif /* DOWNLINKGPUSCHEDULER_H_ */
