/*
 * radiosaber_scheduler.hpp -- C++ host mirror of the reference scheduler's call surface on top of
 * the C ABI (radiosaber_hip.h).  Header-only; link with -lradiosaber_hip.
 *
 * The reference's seam is a class with
 *     DoSchedule()  { UpdateAverageTransmissionRate(); SelectFlowsToSchedule();
 *                     if (users) RBsAllocation(); StopSchedule(); }          (downlink-transport-scheduler.cpp:152-168)
 * operating on LTE-Sim's object graph (RadioBearer, UserToSchedule, PdcchMapIdealControlMessage).
 * This mirror keeps the method names, their order and their arithmetic but holds the bearers as flat
 * records, so it compiles without the LTE-Sim tree; INTEGRATION.md shows the thin subclass that
 * marshals LTE-Sim's objects into it inside the reference tree.
 *
 * Errors follow the reference's behaviour: construction problems throw std::runtime_error (the
 * reference throws on a missing config file, downlink-transport-scheduler.cpp:58-60); per-TTI
 * failures of the GPU path also throw (the reference asserts).  Single-threaded, non-reentrant,
 * like the reference.
 */
#ifndef RADIOSABER_SCHEDULER_HPP_
#define RADIOSABER_SCHEDULER_HPP_

#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <ostream>
#include <vector>

#include "radiosaber_hip.h"

namespace radiosaber {

constexpr int kMaxBearers = 2; /* MAX_BEARERS, packet-scheduler.h:31: a user's bearers are indexed by their priority */

/* RadioBearer fields the path reads and writes (src/flows/radio-bearer.h:81-85) */
struct BearerState {
  double average_transmission_rate = 100000; /* radio-bearer.cpp:54 */
  int transmitted_bytes = 0;
  double last_update = 0;
  unsigned long cumulative_bytes = 0;
  unsigned long cumulative_rbs = 0;
  bool has_packets = true; /* HasPackets(); InfiniteBuffer: always backlogged */
  double hol_delay = 0;    /* GetHeadOfLinePacketDelay(), refreshed by the simulator before DoSchedule() */
  /* one bearer per user (no AddBearer call): what the metric of a customised slice (algo_alpha = 1) sees,
   * m_dataToTransmit[slice_priority_[slice]] != 0.  With AddBearer it is derived from the bearers themselves. */
  bool prio_has_data = true;
  /* bearers of a user (priority 0 exists from the start, application id = user id; AddBearer creates the others) */
  bool exists = false;
  int application_id = -1; /* GetApplication()->GetApplicationID(), the "app:" of the stderr line */
  int queue_size = -1;     /* GetQueueSize() in bytes, refreshed by the simulator; < 0: InfiniteBuffer (100000000) */
};

/* One PDCCH record group: what RBsAllocation() hands to the PHY per scheduled user
 * (downlink-transport-scheduler.cpp:661-668) */
struct Allocation {
  int user_id;
  std::vector<int> prbs; /* GetListOfAllocatedRBs(): RBG-ascending; RS_SCHED_UPPERBOUND: the slice's sorted (push) order */
  int final_cqi, mcs, tbs_bits;
  int n_prbs;            /* size of the reference's list (complete for every scheduler) */
};

class GpuDownlinkScheduler {
 public:
  /* mirrors DownlinkTransportScheduler(config_fname, interslice_algo) after JSON parsing:
   * ues_per_slice + per-slice weight/alpha/beta/epsilon/psi; sched = RS_SCHED_* */
  GpuDownlinkScheduler(const std::vector<int>& ues_per_slice, const std::vector<double>& weight,
                       const std::vector<int>& alpha, const std::vector<int>& beta,
                       const std::vector<int>& epsilon, const std::vector<int>& psi, int nb_rbs, int rbg_size,
                       int sched, int device = 0)
      : num_slices_((int)ues_per_slice.size()), rbg_size_(rbg_size), sched_(sched), weight_(weight),
        alpha_(alpha), beta_(beta), eps_(epsilon), psi_(psi) {
    for (int s = 0; s < num_slices_; ++s)
      for (int j = 0; j < ues_per_slice[s]; ++j) user_to_slice_.push_back(s);
    nb_rbs_ = nb_rbs - (nb_rbs % rbg_size); /* :460 */
    nb_rbgs_ = nb_rbs_ / rbg_size;
    bearers_.resize(user_to_slice_.size() * (size_t)kMaxBearers);
    for (size_t u = 0; u < user_to_slice_.size(); ++u) {
      bearers_[u * kMaxBearers].exists = true;
      bearers_[u * kMaxBearers].application_id = (int)u;
    }
    slice_priority_.assign(num_slices_, 0);
    cqi_.assign(user_to_slice_.size() * (size_t)nb_rbgs_, 10); /* ENodeB.cpp:212-217 */
    slice_ewma_time_.assign(num_slices_, 0.0);
    rs_config c{};
    c.n_slices = num_slices_;
    c.n_users = (int)user_to_slice_.size();
    c.n_rbgs = nb_rbgs_;
    c.rbg_size = rbg_size;
    c.sched = sched;
    c.device = device;
    c.slice_weight = weight_.data();
    c.algo_alpha = alpha_.data();
    c.algo_beta = beta_.data();
    c.algo_epsilon = eps_.data();
    c.algo_psi = psi_.data();
    c.user_to_slice = user_to_slice_.data();
    ctx_ = RS_CREATE(&c); /* rs_create behind the ABI-version + struct-size check */
    if (!ctx_) throw std::runtime_error(std::string("rs_create: ") + rs_last_error());
    /* this context's own build of the one-TTI kernel (hiprtc, ~2 s once per shape and process; identical results, a call
     * ~15-20 % shorter); a failure leaves the context on the kernels built into the library.  RS_DROPIN_JIT=0 skips it. */
    const char* jit = getenv("RS_DROPIN_JIT");
    if (!(jit && jit[0] == '0')) (void)rs_ctx_specialize(ctx_);
  }
  ~GpuDownlinkScheduler() { rs_destroy(ctx_); }
  GpuDownlinkScheduler(const GpuDownlinkScheduler&) = delete;
  GpuDownlinkScheduler& operator=(const GpuDownlinkScheduler&) = delete;

  /* ENodeB::UserEquipmentRecord::SetCQI at RBG granularity (enb-mac-entity.cc:189-191) */
  void SetCQI(int user_id, const uint8_t* cqi_per_rbg) {
    ++cqi_epoch_; /* rs_tti_in.cqi_epoch: the reports changed (the reference: every CQI_INTERVAL = 40 TTIs, enb-mac-entity.cc:38) */
    for (int r = 0; r < nb_rbgs_; ++r) cqi_[(size_t)user_id * nb_rbgs_ + r] = cqi_per_rbg[r];
  }
  /* the full per-PRB vector, as UserEquipmentRecord::GetCQI() holds it (CQIs may differ inside an RBG) */
  void SetCQIPerPrb(int user_id, const int* cqi_per_prb) {
    ++cqi_epoch_;
    if (cqi_prb_.empty()) cqi_prb_.assign(user_to_slice_.size() * (size_t)nb_rbs_, 10);
    for (int k = 0; k < nb_rbs_; ++k) cqi_prb_[(size_t)user_id * nb_rbs_ + k] = (uint8_t)cqi_per_prb[k];
    for (int r = 0; r < nb_rbgs_; ++r) cqi_[(size_t)user_id * nb_rbgs_ + r] = (uint8_t)cqi_per_prb[r * rbg_size_];
  }
  BearerState& Bearer(int user_id, int priority = 0) { return bearers_[(size_t)user_id * kMaxBearers + priority]; }
  /* a further RadioBearer of the user (single-cell-with-interference.h:413-440 gives every UE one InternetFlow per
   * configured priority next to its best-effort flow); priority = RadioBearer::GetPriority() in 0..kMaxBearers-1 */
  BearerState& AddBearer(int user_id, int priority, int application_id) {
    if (priority < 0 || priority >= kMaxBearers) throw std::runtime_error("bearer priority outside 0..MAX_BEARERS-1");
    BearerState& b = Bearer(user_id, priority);
    b.exists = true;
    b.application_id = application_id;
    multi_bearer_ = true;
    return b;
  }
  const std::vector<int>& SlicePriority() const { return slice_priority_; }
  unsigned long GetTimeStamp() const { return ts_; }
  const std::vector<Allocation>& LastAllocations() const { return allocations_; }
  const std::vector<int>& SliceTargetRbs() const { return target_; }
  const std::vector<int>& SliceQuotaRbgs() const { return quota_; }

  /* PacketScheduler::Schedule() -> DoSchedule()  (packet-scheduler.cpp:72-90);
   * now = Simulator::Init()->Now() */
  void DoSchedule(double now) {
    int slice_serve = -1;
    if (sched_ == RS_SCHED_NVS || sched_ == RS_SCHED_NVS_NONGREEDY) slice_serve = SelectSliceToServe(); /* before the EWMA update (nvs :207-208) */
    UpdateAverageTransmissionRate(now);
    SelectFlowsToSchedule(slice_serve);
    allocations_.clear();
    if (!users_.empty()) RBsAllocation();
    DoStopSchedule();
  }

  /* radio-bearer.cpp:139-164 for every bearer (downlink-transport-scheduler.cpp:715-727) */
  void UpdateAverageTransmissionRate(double now) {
    for (BearerState& b : bearers_) {
      if (!b.exists || now == b.last_update) continue;
      double rate = (b.transmitted_bytes * 8) / (now - b.last_update);
      double beta = 0.02;
      b.average_transmission_rate = ((1 - beta) * b.average_transmission_rate) + (beta * rate);
      if (b.average_transmission_rate < 1) b.average_transmission_rate = 1;
      b.transmitted_bytes = 0;
      b.last_update = now;
    }
  }

  /* downlink-nvs-scheduler.cpp:94-142 */
  int SelectSliceToServe() {
    std::vector<bool> with_queue(num_slices_, false);
    for (size_t k = 0; k < bearers_.size(); ++k)
      if (bearers_[k].exists && bearers_[k].has_packets) with_queue[user_to_slice_[k / kMaxBearers]] = true;
    int slice_id = 0;
    double max_score = 0;
    for (int i = 0; i < num_slices_; ++i) {
      if (!with_queue[i]) continue;
      if (slice_ewma_time_[i] == 0) { slice_id = i; break; }
      double score = weight_[i] / slice_ewma_time_[i];
      if (score >= max_score) { max_score = score; slice_id = i; }
    }
    const double beta = 0.01;
    for (int i = 0; i < num_slices_; ++i) {
      if (!with_queue[i]) continue;
      slice_ewma_time_[i] = (1 - beta) * slice_ewma_time_[i];
      if (i == slice_id) slice_ewma_time_[i] += beta * 1;
    }
    return slice_id;
  }

  /* downlink-transport-scheduler.cpp:105-150 (downlink-nvs-scheduler.cpp:144-194 with the slice filter) +
   * PacketScheduler::InsertFlowToUser (packet-scheduler.cpp:305-335): bearers in container order (user, then priority);
   * a bearer with packets enters its user's record at index = priority with dataToTransmit = queue size (100000000 for
   * an InfiniteBuffer) and raises its slice's priority; users in first-seen (= id) order */
  void SelectFlowsToSchedule(int slice_serve = -1) {
    users_.clear();
    data_.clear();
    slice_priority_.assign(num_slices_, 0);
    for (size_t u = 0; u < user_to_slice_.size(); ++u) {
      if (slice_serve >= 0 && user_to_slice_[u] != slice_serve) continue;
      bool seen = false;
      for (int pr = 0; pr < kMaxBearers; ++pr) {
        const BearerState& b = bearers_[u * kMaxBearers + pr];
        if (!b.exists || !b.has_packets) continue;
        if (pr > slice_priority_[user_to_slice_[u]]) slice_priority_[user_to_slice_[u]] = pr;
        if (!seen) {
          users_.push_back((int)u);
          data_.insert(data_.end(), kMaxBearers, -1); /* -1: m_bearers[pr] == NULL */
          seen = true;
        }
        data_[data_.size() - kMaxBearers + pr] = b.queue_size < 0 ? 100000000 : b.queue_size;
      }
    }
  }

  /* downlink-transport-scheduler.cpp:453-675 -- on the GPU through the C ABI.
   * Draws the two rand() values exactly where the reference does (:490, :511). */
  void RBsAllocation() {
    const int n = (int)users_.size();
    in_cqi_.resize((size_t)n * nb_rbgs_);
    in_avg_.resize(n);
    for (int i = 0; i < n; ++i) {
      const int u = users_[i];
      for (int r = 0; r < nb_rbgs_; ++r) in_cqi_[(size_t)i * nb_rbgs_ + r] = cqi_[(size_t)u * nb_rbgs_ + r];
      /* ComputeSchedulingMetric :681-686: averageRate = 1; for every bearer in the record: averageRate += its average.
       * The C ABI adds 1 to one number, so two bearers go in as ((1 + a0) + a1) - 1: a0, a1 >= 1, so the sum K is at least
       * 2, K - 1 is exact in binary64 and 1 + (K - 1) == K -- the reference's summation order, bit for bit. */
      double k = 1;
      int n_present = 0;
      for (int pr = 0; pr < kMaxBearers; ++pr)
        if (data_[(size_t)i * kMaxBearers + pr] >= 0) {
          k += bearers_[(size_t)u * kMaxBearers + pr].average_transmission_rate;
          ++n_present;
        }
      in_avg_[i] = n_present == 1 ? bearers_[(size_t)u * kMaxBearers + (data_[(size_t)i * kMaxBearers] >= 0 ? 0 : 1)].average_transmission_rate
                                  : k - 1;
    }
    if (!cqi_prb_.empty()) {
      in_prb_.resize((size_t)n * nb_rbs_);
      for (int i = 0; i < n; ++i)
        for (int k = 0; k < nb_rbs_; ++k) in_prb_[(size_t)i * nb_rbs_ + k] = cqi_prb_[(size_t)users_[i] * nb_rbs_ + k];
    }
    rs_tti_in in{};
    in.n_users = n;
    in.user_id = users_.data();
    in.cqi = in_cqi_.data();
    in.avg_rate = in_avg_.data();
    in.cqi_prb = cqi_prb_.empty() ? nullptr : in_prb_.data();
    /* between two SetCQI calls the library serves the grid from the image it kept on the device (it compares the user list itself) */
    in.cqi_epoch = cqi_epoch_;
    in_hol_.resize(n);
    in_prio_.resize(n);
    for (int i = 0; i < n; ++i) {
      /* :694-711: the metric of a customised slice looks at the record's entry of the slice priority */
      const int u = users_[i], sp = slice_priority_[user_to_slice_[u]];
      const BearerState& b = bearers_[(size_t)u * kMaxBearers + sp];
      in_hol_[i] = b.hol_delay;
      const int dtt = data_[(size_t)i * kMaxBearers + sp]; /* m_dataToTransmit[slice_priority_[slice]] (0 when absent) */
      in_prio_[i] = multi_bearer_ ? (dtt > 0 ? 1 : 0) : (b.prio_has_data ? 1 : 0);
    }
    in.hol_delay = in_hol_.data();
    in.prio_has_data = in_prio_.data();
    if (sched_ == RS_SCHED_SEQUENTIAL || sched_ == RS_SCHED_MAXCELL || sched_ == RS_SCHED_VOGEL || sched_ == RS_SCHED_SUBOPT || sched_ == RS_SCHED_UPPERBOUND) {
      in.rand0 = rand();
      in.rand1 = rand();
    }
    if (sched_ == RS_SCHED_NVS_NONGREEDY) {
      /* RBsAllocationNonGreedyPF draws 300 x users values from libc rand(), sample-major (nvs :431-441) */
      in_draws_.resize((size_t)300 * n);
      for (int& d : in_draws_) d = rand();
      in.rand_draws = in_draws_.data();
    }
    target_.assign(num_slices_, 0);
    quota_.assign(num_slices_, 0);
    rbg_to_user_.assign(nb_rbgs_, -1);
    nprb_.assign(n, 0); fcqi_.assign(n, 0); mcs_.assign(n, 0); tbs_.assign(n, 0);
    rs_tti_out out{target_.data(), quota_.data(), rbg_to_user_.data(), nprb_.data(), fcqi_.data(), mcs_.data(), tbs_.data(),
                   nullptr, nullptr};
    const bool upper = sched_ == RS_SCHED_UPPERBOUND;
    if (upper) { /* several users may hold one RBG: the per-slice lists carry what the apply step :603-616 walks */
      upper_rbg_.assign((size_t)num_slices_ * nb_rbgs_, -1);
      upper_user_.assign((size_t)num_slices_ * nb_rbgs_, -1);
      out.upper_rbg = upper_rbg_.data();
      out.upper_user = upper_user_.data();
    }
    if (rs_schedule_tti(ctx_, &in, &out) != RS_OK) throw std::runtime_error(std::string("rs_schedule_tti: ") + rs_last_error());
    for (int i = 0; i < n; ++i) {
      if (nprb_[i] == 0) continue;
      Allocation a{users_[i], {}, fcqi_[i], mcs_[i], tbs_[i], nprb_[i]};
      if (upper) { /* push order = the slice's sorted order */
        const int sl = user_to_slice_[users_[i]];
        for (int k = 0; k < nb_rbgs_; ++k)
          if (upper_user_[(size_t)sl * nb_rbgs_ + k] == users_[i]) {
            const int r = upper_rbg_[(size_t)sl * nb_rbgs_ + k];
            for (int q = r * rbg_size_; q < (r + 1) * rbg_size_; ++q) a.prbs.push_back(q);
          }
      } else {
        for (int r = 0; r < nb_rbgs_; ++r)
          if (rbg_to_user_[r] == users_[i])
            for (int k = r * rbg_size_; k < (r + 1) * rbg_size_; ++k) a.prbs.push_back(k);
      }
      allocations_.push_back(a);
    }
    if (log_out_) WriteAllocationLog(*log_out_);
  }

  /* the reference's stdout of RBsAllocation: downlink-transport-scheduler.cpp:523-527 (slice line, transport schedulers
   * only), :631 / downlink-nvs-scheduler.cpp:330 (time stamp), :637-649 / nvs :334-348 (one line per served user).
   * The PF scheduler prints no map. */
  void WriteAllocationLog(std::ostream& os) const {
    if (sched_ == RS_SCHED_PF) return;
    const bool transport = sched_ == RS_SCHED_SEQUENTIAL || sched_ == RS_SCHED_MAXCELL || sched_ == RS_SCHED_VOGEL || sched_ == RS_SCHED_SUBOPT ||
                           sched_ == RS_SCHED_UPPERBOUND;
    if (transport) {
      os << "slice_id, target_rbs, quota_rbgs: ";
      for (int i = 0; i < num_slices_; ++i) os << "(" << i << ", " << target_[i] << ", " << quota_[i] << ") ";
      os << std::endl;
    }
    os << ts_ << std::endl;
    for (const Allocation& a : allocations_) {
      os << "User(" << a.user_id << ") allocated RBGS:";
      for (size_t k = 0; k < a.prbs.size(); k += (size_t)rbg_size_) {
        const int prb = a.prbs[k];
        const int cqi = cqi_prb_.empty() ? cqi_[(size_t)a.user_id * nb_rbgs_ + prb / rbg_size_]
                                         : cqi_prb_[(size_t)a.user_id * nb_rbs_ + prb];
        os << " " << prb / rbg_size_ << "(" << cqi << ")";
      }
      os << " final_cqi: " << a.final_cqi << std::endl;
    }
  }

  /* downlink-transport-scheduler.cpp:170-221 / downlink-nvs-scheduler.cpp:218-272: the granted bytes go to the user's
   * bearers from the highest priority down, each at most its dataToTransmit; every bearer that sends counts the user's
   * whole PRB list.  (The PF scheduler of sched 1 works per flow: one bearer per user there.) */
  void DoStopSchedule() {
    size_t i = 0;
    for (const Allocation& a : allocations_) {
      while (users_[i] != a.user_id) ++i; /* allocations_ follows users_ */
      int available = a.tbs_bits / 8;
      for (int pr = kMaxBearers - 1; pr >= 0; --pr) {
        if (available <= 0) break;
        const int dtt = data_[i * kMaxBearers + pr];
        if (dtt <= 0) continue;
        const int sent = available < dtt ? available : dtt;
        available -= sent;
        BearerState& b = bearers_[(size_t)a.user_id * kMaxBearers + pr];
        b.transmitted_bytes += sent;
        b.cumulative_bytes += sent;
        b.cumulative_rbs += a.n_prbs;
        if (log_err_) {
          /* downlink-transport-scheduler.cpp:192-199, downlink-nvs-scheduler.cpp:240-247 ("app:", with user and slice);
           * downlink-packet-scheduler.cpp:140-145 ("flow:", without) */
          std::ostream& es = *log_err_;
          es << ts_ << (sched_ == RS_SCHED_PF ? " flow: " : " app: ") << b.application_id << " cumu_bytes: "
             << b.cumulative_bytes << " cumu_rbs: " << b.cumulative_rbs << " hol_delay: " << b.hol_delay;
          if (sched_ != RS_SCHED_PF) es << " user: " << a.user_id << " slice: " << user_to_slice_[a.user_id];
          es << std::endl;
        }
      }
    }
    ts_++;
  }

  /* the reference's scheduler counts every TTI from 0 (m_ts); a run that starts scheduling at TTI 100 sets 100 here */
  void SetTimeStamp(unsigned long ts) { ts_ = ts; }
  /* log compatibility with the reference's stdout / stderr (what NSDI23-radiosaber-experiments/ *\/plot_*.py parse); NULL = off */
  void SetLogStreams(std::ostream* out, std::ostream* err) { log_out_ = out; log_err_ = err; }

 private:
  int num_slices_, rbg_size_, sched_, nb_rbs_ = 0, nb_rbgs_ = 0;
  std::vector<double> weight_;
  std::vector<int> alpha_, beta_, eps_, psi_, user_to_slice_;
  std::vector<double> slice_ewma_time_;
  std::vector<BearerState> bearers_; /* [user][priority] */
  std::vector<uint8_t> cqi_, in_cqi_, cqi_prb_, in_prb_;
  uint64_t cqi_epoch_ = 1; /* version number of cqi_ / cqi_prb_ (never 0: 0 means "no promise") */
  std::vector<double> in_avg_, in_hol_;
  std::vector<uint8_t> in_prio_;
  std::vector<int> in_draws_, upper_rbg_, upper_user_;
  std::vector<int> users_, target_, quota_, rbg_to_user_, nprb_, fcqi_, mcs_, tbs_;
  std::vector<int> data_;           /* [users_][kMaxBearers] m_dataToTransmit of the record, -1: no such bearer this TTI */
  std::vector<int> slice_priority_; /* slice_priority_ (downlink-transport-scheduler.h:40) */
  bool multi_bearer_ = false;
  std::vector<Allocation> allocations_;
  unsigned long ts_ = 0;
  std::ostream *log_out_ = nullptr, *log_err_ = nullptr;
  rs_ctx* ctx_ = nullptr;
};

}  // namespace radiosaber

#endif /* RADIOSABER_SCHEDULER_HPP_ */
