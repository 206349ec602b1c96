/* dl-gpu-pf-packet-scheduler.h -- DL_PF_PacketScheduler (CLI scheduler 1, "No-Slicing PF") with RBsAllocation() on an MI355X.
 * Copy next to dl-pf-packet-scheduler.h.  Needs the reference tree's generated load-parameters.h and jsoncpp to compile.
 * The metric, SelectFlowsToSchedule, DoStopSchedule and the EWMA stay the parent's; only
 * DownlinkPacketScheduler::RBsAllocation (downlink-packet-scheduler.cpp:179-331) is replaced. */
#ifndef DLGPUPFPACKETSCHEDULER_H_
#define DLGPUPFPACKETSCHEDULER_H_

#include <string>
#include <vector>

#include "dl-pf-packet-scheduler.h"
#include "radiosaber_hip.h"

class DL_GPU_PF_PacketScheduler : public DL_PF_PacketScheduler {
 public:
  /* max_flows: upper bound on simultaneously scheduled flows (bearers with packets), at most RS_MAX_USERS */
  DL_GPU_PF_PacketScheduler(std::string config_fname, int max_flows = RS_MAX_USERS, int hip_device = 0);
  virtual ~DL_GPU_PF_PacketScheduler();

  virtual void RBsAllocation();

 private:
  rs_ctx* ctx_;
  int hip_device_, max_flows_, nb_rbs_;
  std::vector<uint8_t> cqi_prb_;    /* the previous call's per-PRB reports (rs_tti_in.cqi_epoch) */
  unsigned long long cqi_epoch_;    /* bumped when any report changed */
};

#endif /* DLGPUPFPACKETSCHEDULER_H_ */
