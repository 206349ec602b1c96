/* downlink-gpu-scheduler.cpp -- see downlink-gpu-scheduler.h.  Include list = the parent's (downlink-transport-scheduler.cpp:22-47). */
#include "downlink-gpu-scheduler.h"

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
int SchedOfInterSliceAlgo(int algo) {
  switch (algo) {
    case 0: return RS_SCHED_SEQUENTIAL;
    case 1: return RS_SCHED_SUBOPT;
    case 2: return RS_SCHED_MAXCELL;
    case 3: return RS_SCHED_VOGEL;
    case 4: return RS_SCHED_UPPERBOUND;
    default: throw std::runtime_error("DownlinkGpuScheduler: unknown inter-slice algorithm");
  }
}
}  // namespace

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

DownlinkGpuScheduler::DownlinkGpuScheduler(std::string config_fname, int interslice_algo, int hip_device)
    : DownlinkTransportScheduler(config_fname, interslice_algo),
      ctx_(NULL), hip_device_(GpuProcessPolicy(hip_device)), sched_(SchedOfInterSliceAlgo(interslice_algo)), num_slices_(0),
      nb_rbs_(0), rbg_size_(0), any_alpha_(false), cqi_epoch_(0) {
  /* the same keys the parent's constructor reads (downlink-transport-scheduler.cpp:55-88) */
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
      any_alpha_ = any_alpha_ || alpha_.back() != 0;
    }
}

DownlinkGpuScheduler::~DownlinkGpuScheduler() { rs_destroy(ctx_); }

void DownlinkGpuScheduler::LazyCreate(int nb_rbs, int rbg_size) {
  /* the PRB grid is only known once the MAC entity is attached (SetMacEntity runs after the constructor, ENodeB.cpp:303-391) */
  rs_config cfg;
  cfg.n_slices = num_slices_;
  cfg.n_users = (int)user_to_slice_.size();
  cfg.n_rbgs = nb_rbs / rbg_size;
  cfg.rbg_size = rbg_size;
  cfg.sched = sched_;
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
  cfg.synthetic_exp = 1; /* the parent's transport block, :653-659 */
#endif
  cfg.link_tables = RS_LINK_HOST_LIBM; /* the EESM constants of THIS machine's libm, like the CPU schedulers linked into the same binary */
  ctx_ = RS_CREATE(&cfg);
  if (!ctx_) throw std::runtime_error(std::string("rs_create: ") + rs_last_error());
  (void)rs_ctx_specialize(ctx_); /* this shape's own build of the one-TTI kernel (~2 s at start-up; on failure the built-in kernels stay) */
  nb_rbs_ = nb_rbs;
  rbg_size_ = rbg_size;
}

void DownlinkGpuScheduler::RBsAllocation() {
  UsersToSchedule* users = GetUsersToSchedule(); /* packet-scheduler.h:88-123: users with queued data, first-seen order */
  int nb_rbs = GetMacEntity()->GetDevice()->GetPhy()->GetBandwidthManager()->GetDlSubChannels().size();
  const int rbg_size = rs_get_rbg_size(nb_rbs); /* = get_rbg_size(), utility/eesm-effective-sinr.h:82-103 */
  if (rbg_size < 0) throw std::runtime_error(rs_last_error());
  nb_rbs -= nb_rbs % rbg_size; /* :460 */
  const int R = nb_rbs / rbg_size;
  if (!ctx_) LazyCreate(nb_rbs, rbg_size);
  if (nb_rbs != nb_rbs_) throw std::runtime_error("DownlinkGpuScheduler: the PRB grid changed after the first TTI");

  const int n = (int)users->size();
  std::vector<int> ids(n);
  /* the per-PRB reports live in a member: a UE's CQI changes every CQI_INTERVAL = 40 TTIs (enb-mac-entity.cc:38, cqi-manager.cpp:115), so
   * the gather below notices for free whether ANY byte differs from the previous TTI's and bumps rs_tti_in.cqi_epoch only then -- the
   * library then serves 39 calls of 40 from the image it kept on the device (it checks the user list itself) */
  bool cqi_changed = cqi_prb_.size() != (size_t)n * nb_rbs;
  cqi_prb_.resize((size_t)n * nb_rbs);
  std::vector<uint8_t>& cqi_prb = cqi_prb_;
  std::vector<double> avg(n), hol(n, 0.0);
  std::vector<uint8_t> prio_has_data(n, 1);
  /* slice_priority_ is private in the parent: highest priority among the bearers that have packets, per slice
   * (SelectFlowsToSchedule :115-146 inserts exactly those bearers into m_bearers[priority]) */
  std::vector<int> slice_priority(num_slices_, 0);
  for (int i = 0; i < n; i++) {
    UserToSchedule* u = users->at(i);
    const int sid = user_to_slice_[u->GetUserID()];
    for (int b = 0; b < MAX_BEARERS; b++)
      if (u->m_bearers[b] && b > slice_priority[sid]) slice_priority[sid] = b;
  }
  for (int i = 0; i < n; i++) {
    UserToSchedule* u = users->at(i);
    ids[i] = u->GetUserID();
    if (i && ids[i] <= ids[i - 1]) throw std::runtime_error("DownlinkGpuScheduler: users are not in ascending id order");
    /* the full per-PRB report: the metric reads PRB rbg*rbg_size (:536), link adaptation every allocated PRB (:643-646) */
    const std::vector<int>& fb = u->GetCqiFeedbacks();
    for (int k = 0; k < nb_rbs; k++) {
      const uint8_t v = (uint8_t)fb.at(k);
      uint8_t& slot = cqi_prb[(size_t)i * nb_rbs + k];
      cqi_changed |= slot != v;
      slot = v;
    }
    /* ComputeSchedulingMetric :681-686: averageRate = 1; += every bearer's average.  The device forms 1 + avg[i];
     * ((1 + a0) + a1) - 1 is exact for averages >= 1, so two bearers keep the reference's summation order bit for bit */
    double k1 = 1, only = 0;
    int nb = 0;
    for (int b = 0; b < MAX_BEARERS; b++)
      if (u->m_bearers[b]) {
        only = u->m_bearers[b]->GetAverageTransmissionRate();
        k1 += only;
        nb++;
      }
    avg[i] = nb == 1 ? only : k1 - 1;
    const int sid = user_to_slice_[ids[i]];
    if (alpha_[sid]) { /* customised slice, :694-711 */
      const int p = slice_priority[sid];
      prio_has_data[i] = u->m_dataToTransmit[p] != 0;
      if (beta_[sid] && u->m_bearers[p]) hol[i] = u->m_bearers[p]->GetHeadOfLinePacketDelay();
    }
  }

  rs_tti_in in;
  in.n_users = n;
  in.user_id = ids.data();
  in.cqi = NULL;
  in.avg_rate = avg.data();
  in.rand0 = rand(); /* the draw of :490 ... */
  in.rand1 = rand(); /* ... and of :511, in this order: the libc stream stays the simulator's */
  in.cqi_prb = cqi_prb.data();
  in.hol_delay = any_alpha_ ? hol.data() : NULL;
  in.prio_has_data = any_alpha_ ? prio_has_data.data() : NULL;
  in.rand_draws = NULL;
  in.required_rbs = NULL; /* DownlinkTransportScheduler::RBsAllocation has no per-user gate */
  in.data_to_transmit = NULL;
  if (cqi_changed) ++cqi_epoch_;
  in.cqi_epoch = cqi_epoch_; /* (never 0: starts at 1) */

  std::vector<int> target(num_slices_), quota(num_slices_), map(R), nprb(n), fcqi(n), mcs(n), tbs(n);
  std::vector<int> upper_rbg, upper_user;
  rs_tti_out out;
  out.target_rbs = target.data();
  out.quota_rbgs = quota.data();
  out.rbg_to_user = map.data();
  out.user_nprb = nprb.data();
  out.user_final_cqi = fcqi.data();
  out.user_mcs = mcs.data();
  out.user_tbs_bits = tbs.data();
  out.upper_rbg = out.upper_user = NULL;
  if (sched_ == RS_SCHED_UPPERBOUND) {
    upper_rbg.assign((size_t)num_slices_ * R, -1);
    upper_user.assign((size_t)num_slices_ * R, -1);
    out.upper_rbg = upper_rbg.data();
    out.upper_user = upper_user.data();
  }
  if (rs_schedule_tti(ctx_, &in, &out) != RS_OK) throw std::runtime_error(std::string("rs_schedule_tti: ") + rs_last_error());

  /* what the reference's RBsAllocation leaves behind (:589-674): allocation lists, allocated bits, PDCCH records, stdout */
  std::cout << "slice_id, target_rbs, quota_rbgs: ";
  for (int s = 0; s < num_slices_; s++) std::cout << "(" << s << ", " << target[s] << ", " << quota[s] << ") ";
  std::cout << std::endl;
  if (sched_ == RS_SCHED_UPPERBOUND) {
    /* several slices may hold one RBG (:603-616): every slice's RBGs in its push order */
    for (int s = 0; s < num_slices_; s++)
      for (int k = 0; k < R && upper_rbg[(size_t)s * R + k] >= 0; k++) {
        const int rbg = upper_rbg[(size_t)s * R + k], uid = upper_user[(size_t)s * R + k];
        for (int i = 0; i < n; i++)
          if (ids[i] == uid)
            for (int j = rbg * rbg_size; j < (rbg + 1) * rbg_size; j++) users->at(i)->GetListOfAllocatedRBs()->push_back(j);
      }
  } else {
    for (int r = 0; r < R; r++) {
      if (map[r] < 0) continue;
      for (int i = 0; i < n; i++)
        if (ids[i] == map[r])
          for (int j = r * rbg_size; j < (r + 1) * rbg_size; j++) users->at(i)->GetListOfAllocatedRBs()->push_back(j);
    }
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
