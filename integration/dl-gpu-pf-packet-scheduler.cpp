/* dl-gpu-pf-packet-scheduler.cpp -- see dl-gpu-pf-packet-scheduler.h. */
#include <cstdlib>
#include "dl-gpu-pf-packet-scheduler.h"

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

DL_GPU_PF_PacketScheduler::DL_GPU_PF_PacketScheduler(std::string config_fname, int max_flows, int hip_device)
    : DL_PF_PacketScheduler(config_fname), ctx_(NULL), hip_device_(GpuProcessPolicy(hip_device)), max_flows_(max_flows), nb_rbs_(0), cqi_epoch_(0) {}

DL_GPU_PF_PacketScheduler::~DL_GPU_PF_PacketScheduler() { rs_destroy(ctx_); }

void DL_GPU_PF_PacketScheduler::RBsAllocation() {
  FlowsToSchedule* flows = GetFlowsToSchedule(); /* one record per bearer with packets, RRC container order */
  const int nb_rbs = GetMacEntity()->GetDevice()->GetPhy()->GetBandwidthManager()->GetDlSubChannels().size();
  const int rbg_size = rs_get_rbg_size(nb_rbs);
  if (rbg_size < 0) throw std::runtime_error(rs_last_error());
  /* the reference rounds the RBG count UP here (:190) and clips the last RBG; the C ABI takes whole RBGs only */
  if (nb_rbs % rbg_size) throw std::runtime_error("DL_GPU_PF_PacketScheduler: nb_rbs is not a multiple of the RBG size");
  const int R = nb_rbs / rbg_size;
  const int n = (int)flows->size();
  if (n > max_flows_) throw std::runtime_error("DL_GPU_PF_PacketScheduler: more flows than max_flows");
  if (!ctx_) {
    /* no slices in this scheduler: one slice holding every flow slot */
    static const double kWeight = 1.0;
    static const int kZero = 0, kOne = 1;
    std::vector<int> u2s(max_flows_, 0);
    rs_config cfg;
    cfg.n_slices = 1;
    cfg.n_users = max_flows_;
    cfg.n_rbgs = R;
    cfg.rbg_size = rbg_size;
    cfg.sched = RS_SCHED_PF;
    cfg.device = hip_device_;
    cfg.slice_weight = &kWeight;
    cfg.algo_alpha = &kZero;
    cfg.algo_beta = &kZero;
    cfg.algo_epsilon = &kOne;
    cfg.algo_psi = &kOne;
    cfg.user_to_slice = u2s.data();
    cfg.stream = NULL;
    cfg.synthetic_exp = 0; /* DownlinkPacketScheduler::RBsAllocation has no synthetic-experiment branch */
    cfg.link_tables = RS_LINK_HOST_LIBM; /* this machine's libm, like the CPU schedulers in the same binary */
    ctx_ = RS_CREATE(&cfg);
    if (!ctx_) throw std::runtime_error(std::string("rs_create: ") + rs_last_error());
    (void)rs_ctx_specialize(ctx_); /* this shape's own build of the one-TTI kernel (~2 s at start-up; on failure the built-in kernels stay) */
    nb_rbs_ = nb_rbs;
  }
  if (nb_rbs != nb_rbs_) throw std::runtime_error("DL_GPU_PF_PacketScheduler: the PRB grid changed after the first TTI");

  /* the "users" of the C ABI are this scheduler's flows, ids = positions in the flow list (user_id NULL) */
  /* (rs_tti_in.cqi_epoch: a flow's reports change every 40 TTIs; with the same flows and the same reports as the call before, the
   * library schedules from the image it kept on the device) */
  bool cqi_changed = cqi_prb_.size() != (size_t)n * nb_rbs;
  cqi_prb_.resize((size_t)n * nb_rbs);
  std::vector<uint8_t>& cqi_prb = cqi_prb_;
  std::vector<double> avg(n);
  std::vector<int> data(n);
  for (int i = 0; i < n; i++) {
    FlowToSchedule* f = flows->at(i);
    /* the break at :253-265 (transport block >= dataToTransmit * 8): rs_tti_in.data_to_transmit carries every flow's queue */
    data[i] = f->GetDataToTransmit();
    const std::vector<int> fb = f->GetCqiFeedbacks();
    for (int k = 0; k < nb_rbs; k++) {
      const uint8_t v = (uint8_t)fb.at(k);
      uint8_t& slot = cqi_prb[(size_t)i * nb_rbs + k];
      cqi_changed |= slot != v;
      slot = v;
    }
    avg[i] = f->GetBearer()->GetAverageTransmissionRate(); /* metric (se * 180000.) / avg, dl-pf-packet-scheduler.cpp:128-140 */
  }
  rs_tti_in in;
  in.n_users = n;
  in.user_id = NULL;
  in.cqi = NULL;
  in.avg_rate = avg.data();
  in.rand0 = in.rand1 = 0; /* this scheduler draws nothing */
  in.cqi_prb = cqi_prb.data();
  in.hol_delay = NULL;
  in.prio_has_data = NULL;
  in.rand_draws = NULL;
  in.required_rbs = NULL;
  in.data_to_transmit = data.data();
  if (cqi_changed) ++cqi_epoch_;
  in.cqi_epoch = cqi_epoch_;
  int target = 0, quota = 0;
  std::vector<int> map(R), nprb(n), fcqi(n), mcs(n), tbs(n);
  rs_tti_out out;
  out.target_rbs = &target;
  out.quota_rbgs = &quota;
  out.rbg_to_user = map.data();
  out.user_nprb = nprb.data();
  out.user_final_cqi = fcqi.data();
  out.user_mcs = mcs.data();
  out.user_tbs_bits = tbs.data();
  out.upper_rbg = out.upper_user = NULL;
  if (rs_schedule_tti(ctx_, &in, &out) != RS_OK) throw std::runtime_error(std::string("rs_schedule_tti: ") + rs_last_error());

  /* :268-322 -- allocation lists, allocated bits, PDCCH records */
  for (int r = 0; r < R; r++)
    if (map[r] >= 0)
      for (int j = r * rbg_size; j < (r + 1) * rbg_size; j++) flows->at(map[r])->GetListOfAllocatedRBs()->push_back(j);
  PdcchMapIdealControlMessage* pdcchMsg = new PdcchMapIdealControlMessage();
  for (int i = 0; i < n; i++) {
    FlowToSchedule* flow = flows->at(i);
    if (flow->GetListOfAllocatedRBs()->size() == 0) continue;
    flow->UpdateAllocatedBits(tbs[i]);
    for (size_t rb = 0; rb < flow->GetListOfAllocatedRBs()->size(); rb++)
      pdcchMsg->AddNewRecord(PdcchMapIdealControlMessage::DOWNLINK, flow->GetListOfAllocatedRBs()->at(rb),
                             flow->GetBearer()->GetDestination(), mcs[i]);
  }
  if (pdcchMsg->GetMessage()->size() > 0) GetMacEntity()->GetDevice()->GetPhy()->SendIdealControlMessage(pdcchMsg);
  delete pdcchMsg;
}
