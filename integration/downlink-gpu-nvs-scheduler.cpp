/* downlink-gpu-nvs-scheduler.cpp -- see downlink-gpu-nvs-scheduler.h. */
#include "downlink-gpu-nvs-scheduler.h"

#include <jsoncpp/json/json.h>

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>

#include "../../../core/spectrum/bandwidth-manager.h"
#include "../../../device/ENodeB.h"
#include "../../../device/NetworkNode.h"
#include "../../../flows/radio-bearer.h"
#include "../../../phy/lte-phy.h"
#include "../mac-entity.h"

namespace {
/* One simulator process = one GPU context = one HIP stream: ask the HIP runtime for ONE hardware queue instead of its default of four
 * (GPU_MAX_HW_QUEUES, read at the runtime's first call) -- a dozen such processes then share an MI355X without oversubscribing its
 * hardware queue slots, with the default six of them already stall each other for milliseconds (profiles/r06_dropin_concurrency.md) --
 * and pick the GPU: RS_HIP_DEVICE when set (the experiment scripts start one process per run: export RS_HIP_DEVICE=$((i % 8)) in their loops), else the
 * constructor's argument.  The environment wins over both. */
int GpuProcessPolicy(int hip_device) {
  setenv("GPU_MAX_HW_QUEUES", "1", 0 /* keep what the user exported */);
  const char* d = getenv("RS_HIP_DEVICE");
  return d ? atoi(d) : hip_device;
}
}  // namespace

DownlinkGpuNVSScheduler::DownlinkGpuNVSScheduler(std::string config_fname, bool is_nongreedy, int hip_device)
    : DownlinkNVSScheduler(config_fname, is_nongreedy), ctx_(NULL), hip_device_(GpuProcessPolicy(hip_device)), nongreedy_(is_nongreedy),
      num_slices_(0), nb_rbs_(0), cqi_epoch_(0) {
  /* the keys the parent's constructor reads (downlink-nvs-scheduler.cpp:44-86); its members are private */
  std::ifstream ifs(config_fname);
  if (!ifs.is_open()) throw std::runtime_error("Fail to open configuration file.");
  Json::Reader reader;
  Json::Value obj;
  reader.parse(ifs, obj);
  ifs.close();
  const Json::Value& ues_per_slice = obj["ues_per_slice"];
  num_slices_ = ues_per_slice.size();
  for (int i = 0; i < num_slices_; i++)
    for (int j = 0; j < ues_per_slice[i].asInt(); j++) user_to_slice_.push_back(i);
  const Json::Value& slice_schemes = obj["slices"];
  for (unsigned i = 0; i < slice_schemes.size(); i++)
    for (int j = 0; j < slice_schemes[i]["n_slices"].asInt(); j++) {
      slice_weights_.push_back(slice_schemes[i]["weight"].asDouble());
      alpha_.push_back(slice_schemes[i]["algo_alpha"].asInt());
      beta_.push_back(slice_schemes[i]["algo_beta"].asInt());
      epsilon_.push_back(slice_schemes[i]["algo_epsilon"].asInt());
      psi_.push_back(slice_schemes[i]["algo_psi"].asInt());
    }
}

DownlinkGpuNVSScheduler::~DownlinkGpuNVSScheduler() { rs_destroy(ctx_); }

void DownlinkGpuNVSScheduler::DoSchedule(void) {
  /* downlink-nvs-scheduler.cpp:196-218, with this class's RBsAllocation for both flavours */
  int slice_serve = SelectSliceToServe();
  UpdateAverageTransmissionRate(slice_serve);
  SelectFlowsToSchedule(slice_serve);
  if (GetUsersToSchedule()->size() != 0) RBsAllocation();
  StopSchedule();
}

void DownlinkGpuNVSScheduler::RBsAllocation() {
  UsersToSchedule* users = GetUsersToSchedule(); /* the users of the served slice only (SelectFlowsToSchedule(int) :144-194) */
  int nb_rbs = GetMacEntity()->GetDevice()->GetPhy()->GetBandwidthManager()->GetDlSubChannels().size();
  const int rbg_size = rs_get_rbg_size(nb_rbs);
  if (rbg_size < 0) throw std::runtime_error(rs_last_error());
  nb_rbs -= nb_rbs % rbg_size;
  const int R = nb_rbs / rbg_size;
  if (!ctx_) {
    rs_config cfg;
    cfg.n_slices = num_slices_;
    cfg.n_users = (int)user_to_slice_.size();
    cfg.n_rbgs = R;
    cfg.rbg_size = rbg_size;
    cfg.sched = nongreedy_ ? RS_SCHED_NVS_NONGREEDY : RS_SCHED_NVS;
    cfg.device = hip_device_;
    cfg.slice_weight = slice_weights_.data();
    cfg.algo_alpha = alpha_.data();
    cfg.algo_beta = beta_.data();
    cfg.algo_epsilon = epsilon_.data();
    cfg.algo_psi = psi_.data();
    cfg.user_to_slice = user_to_slice_.data();
    cfg.stream = NULL;
    cfg.synthetic_exp = 0;
#if defined(FIRST_SYNTHETIC_EXP) || defined(SECOND_SYNTHETIC_EXP)
    cfg.synthetic_exp = 1; /* downlink-nvs-scheduler.cpp:336-342 (the sampler of :405-528 has no such branch: ignored there) */
#endif
    cfg.link_tables = RS_LINK_HOST_LIBM; /* this machine's libm, like the CPU schedulers in the same binary */
    ctx_ = RS_CREATE(&cfg);
    if (!ctx_) throw std::runtime_error(std::string("rs_create: ") + rs_last_error());
    (void)rs_ctx_specialize(ctx_); /* this shape's own build of the one-TTI kernel (~2 s at start-up; on failure the built-in kernels stay) */
    nb_rbs_ = nb_rbs;
  }
  if (nb_rbs != nb_rbs_) throw std::runtime_error("DownlinkGpuNVSScheduler: the PRB grid changed after the first TTI");

  const int n = (int)users->size();
  const int sid = user_to_slice_[users->at(0)->GetUserID()];
  int slice_priority = 0; /* the parent's slice_priority_ is private: highest priority among the inserted bearers */
  for (int i = 0; i < n; i++)
    for (int b = 0; b < MAX_BEARERS; b++)
      if (users->at(i)->m_bearers[b] && b > slice_priority) slice_priority = b;
  std::vector<int> ids(n);
  /* (rs_tti_in.cqi_epoch: the reports of the previous call stay in a member; a slice served twice in a row with unchanged reports
   * is scheduled from the context's device-resident image -- the library compares the user list itself) */
  bool cqi_changed = cqi_prb_.size() != (size_t)n * nb_rbs;
  cqi_prb_.resize((size_t)n * nb_rbs);
  std::vector<uint8_t>& cqi_prb = cqi_prb_;
  std::vector<uint8_t> prio_has_data(n, 1);
  std::vector<double> avg(n), hol(n, 0.0);
  std::vector<int> required(n);
  for (int i = 0; i < n; i++) {
    UserToSchedule* u = users->at(i);
    ids[i] = u->GetUserID();
    if (i && ids[i] <= ids[i - 1]) throw std::runtime_error("DownlinkGpuNVSScheduler: users are not in ascending id order");
    /* the gate at :299-300 (allocated PRBs < m_requiredRBs): rs_tti_in.required_rbs */
    required[i] = u->m_requiredRBs;
    const std::vector<int>& fb = u->GetCqiFeedbacks();
    for (int k = 0; k < nb_rbs; k++) {
      const uint8_t v = (uint8_t)fb.at(k);
      uint8_t& slot = cqi_prb[(size_t)i * nb_rbs + k];
      cqi_changed |= slot != v;
      slot = v;
    }
    double k1 = 1, only = 0; /* :364-369: averageRate = 1; += every bearer's average */
    int nb = 0;
    for (int b = 0; b < MAX_BEARERS; b++)
      if (u->m_bearers[b]) {
        only = u->m_bearers[b]->GetAverageTransmissionRate();
        k1 += only;
        nb++;
      }
    avg[i] = nb == 1 ? only : k1 - 1;
    if (alpha_[sid]) { /* :375-387: 0 without prioritized data, else HoL * ratio (always the HoL here) */
      prio_has_data[i] = u->m_dataToTransmit[slice_priority] != 0;
      if (u->m_bearers[slice_priority]) hol[i] = u->m_bearers[slice_priority]->GetHeadOfLinePacketDelay();
    }
  }
  std::vector<int> draws;
  if (nongreedy_) {
    /* RBsAllocationNonGreedyPF :431-441 draws rand() once per user per sample, sample-major: same order here */
    draws.resize((size_t)300 * n);
    for (size_t k = 0; k < draws.size(); k++) draws[k] = rand();
  }
  rs_tti_in in;
  in.n_users = n;
  in.user_id = ids.data();
  in.cqi = NULL;
  in.avg_rate = avg.data();
  in.rand0 = in.rand1 = 0;
  in.cqi_prb = cqi_prb.data();
  in.hol_delay = alpha_[sid] ? hol.data() : NULL;
  in.prio_has_data = alpha_[sid] ? prio_has_data.data() : NULL;
  in.rand_draws = nongreedy_ ? draws.data() : NULL;
  in.required_rbs = nongreedy_ ? NULL : required.data(); /* RBsAllocationNonGreedyPF has no such gate */
  in.data_to_transmit = NULL;
  if (cqi_changed) ++cqi_epoch_;
  in.cqi_epoch = cqi_epoch_;
  std::vector<int> target(num_slices_), quota(num_slices_), map(R), nprb(n), fcqi(n), mcs(n), tbs(n);
  rs_tti_out out;
  out.target_rbs = target.data();
  out.quota_rbgs = quota.data();
  out.rbg_to_user = map.data();
  out.user_nprb = nprb.data();
  out.user_final_cqi = fcqi.data();
  out.user_mcs = mcs.data();
  out.user_tbs_bits = tbs.data();
  out.upper_rbg = out.upper_user = NULL;
  if (rs_schedule_tti(ctx_, &in, &out) != RS_OK) throw std::runtime_error(std::string("rs_schedule_tti: ") + rs_last_error());

  for (int r = 0; r < R; r++) {
    if (map[r] < 0) continue;
    for (int i = 0; i < n; i++)
      if (ids[i] == map[r])
        for (int j = r * rbg_size; j < (r + 1) * rbg_size; j++) users->at(i)->GetListOfAllocatedRBs()->push_back(j);
  }
  PdcchMapIdealControlMessage* pdcchMsg = new PdcchMapIdealControlMessage();
  std::cout << GetTimeStamp() << std::endl;
  for (int i = 0; i < n; i++) {
    UserToSchedule* ue = users->at(i);
    if (ue->GetListOfAllocatedRBs()->size() == 0) continue;
    std::cout << "User(" << ue->GetUserID() << ") allocated RBGS:";
    for (size_t k = 0; k < ue->GetListOfAllocatedRBs()->size(); k++) {
      const int rbid = ue->GetListOfAllocatedRBs()->at(k);
      if (rbid % rbg_size == 0) std::cout << " " << rbid / rbg_size << "(" << ue->GetCqiFeedbacks().at(rbid) << ")";
    }
    std::cout << " final_cqi: " << fcqi[i] << std::endl;
    ue->UpdateAllocatedBits(tbs[i]);
    for (size_t rb = 0; rb < ue->GetListOfAllocatedRBs()->size(); rb++)
      pdcchMsg->AddNewRecord(PdcchMapIdealControlMessage::DOWNLINK, ue->GetListOfAllocatedRBs()->at(rb), ue->GetUserNode(), mcs[i]);
  }
  if (pdcchMsg->GetMessage()->size() > 0) GetMacEntity()->GetDevice()->GetPhy()->SendIdealControlMessage(pdcchMsg);
  delete pdcchMsg;
}
