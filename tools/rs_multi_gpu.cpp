/*
 * rs_multi_gpu.cpp -- the multi-GPU form of BASELINE configs[4] with a C++ host and RCCL called directly (VERDICT r02 next #7;
 * north_star: "the host stays C++ ... RCCL over xGMI only for the final throughput aggregation").
 *
 * ONE process, one rs_batch per visible GPU (rs_config.device), independent cells sharded over the GPUs exactly as bench.py
 * shards them over ranks (global cell id g: GPU g / cells_per_gpu; srand() seed and synthetic CQI grids are functions of g, so a
 * cell's trajectory does not depend on the number of GPUs), rs_batch_run_async on every batch, no data-path collective; at the
 * end every batch reduces its per-slice cumulative bytes on its own device (rs_batch_slice_bytes_device) and ONE
 * ncclAllReduce(ncclUint64, ncclSum) over the ncclCommInitAll communicators sums them -- exact integers.  The per-slice rate it
 * prints is the reference's post-hoc metric (NSDI23-radiosaber-experiments/exp-customization/plot_throughput.py:26-56:
 * cumulative bytes per slice * 8 / 1e6 / simulated seconds; run_backlogged.sh:6-14 starts the independent (scheduler, seed)
 * processes this program's cells stand for).
 *
 *   hipcc -O2 -std=c++17 tools/rs_multi_gpu.cpp -Iinclude -Lradiosaber_amd -lradiosaber_hip -lrccl \
 *         -Wl,-rpath,'$ORIGIN/../radiosaber_amd' -o tools/rs_multi_gpu            (tools/build_multi_gpu.sh)
 *   tools/rs_multi_gpu --gpus 8 --cells 512 --ttis 8000 --launches 20            (configs[4]: 4 096 cells on 8 x MI355X)
 *   tools/rs_multi_gpu --gpus 1 --cells 8 --ttis 200 --launches 2 --check        (what tests/test_gpu_round3.py runs)
 *
 * --check: the all-reduced vector must equal the sum of the batches' host-side rs_batch_slice_bytes and be identical on every
 * GPU; exit code 1 otherwise.  Prints one JSON line.
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "radiosaber_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 2; } } while (0)
#define RS_OK_(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: %s\n", #x, rs_last_error()); return 2; } } while (0)

/* the trace corpus' CQI histogram, the same weights bench.py passes */
static const double kHist[15] = {152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
                                 6890232, 4770864, 2842552, 3579624, 96000, 1227696};

int main(int argc, char** argv) {
  int gpus = 0, cells = 512, ttis = 8000, launches = 5, S = 20, ups = 25, R = 25, G = 4, sched = RS_SCHED_MAXCELL, check = 0;
  for (int i = 1; i < argc; ++i) {
    auto arg = [&](const char* name, int* dst) {
      if (!strcmp(argv[i], name) && i + 1 < argc) { *dst = atoi(argv[++i]); return true; }
      return false;
    };
    if (arg("--gpus", &gpus) || arg("--cells", &cells) || arg("--ttis", &ttis) || arg("--launches", &launches) ||
        arg("--slices", &S) || arg("--ues-per-slice", &ups) || arg("--rbgs", &R) || arg("--rbg-size", &G) || arg("--sched", &sched))
      continue;
    if (!strcmp(argv[i], "--check")) { check = 1; continue; }
    fprintf(stderr, "usage: %s [--gpus N] [--cells per-GPU] [--ttis per-launch] [--launches K] [--slices S] [--ues-per-slice n] "
                    "[--rbgs R] [--rbg-size G] [--sched id] [--check]\n", argv[0]);
    return 2;
  }
  const int visible = rs_device_count();
  if (visible < 1) { fprintf(stderr, "no HIP device: the product has no CPU path\n"); return 2; }
  if (gpus <= 0 || gpus > visible) gpus = visible;
  const int U = S * ups;
  std::vector<double> weight(S, 1.0 / S);
  std::vector<int32_t> zeros(S, 0), ones(S, 1), u2s(U);
  for (int u = 0; u < U; ++u) u2s[u] = u / ups;

  std::vector<rs_batch*> batch(gpus, nullptr);
  std::vector<uint64_t*> d_vec(gpus, nullptr);
  const int kWarm = 40; /* one untimed CQI epoch: code object load, first touch */
  const int n_epochs = (launches * ttis + kWarm + 39) / 40;
  for (int d = 0; d < gpus; ++d) {
    rs_batch_config bc;
    memset(&bc, 0, sizeof bc);
    bc.cell.n_slices = S; bc.cell.n_users = U; bc.cell.n_rbgs = R; bc.cell.rbg_size = G; bc.cell.sched = sched; bc.cell.device = d;
    bc.cell.slice_weight = weight.data(); bc.cell.algo_alpha = zeros.data(); bc.cell.algo_beta = zeros.data();
    bc.cell.algo_epsilon = ones.data(); bc.cell.algo_psi = ones.data(); bc.cell.user_to_slice = u2s.data();
    bc.n_cells = cells; bc.first_tti = 100; bc.cqi_refresh = 40; bc.phy_error_draws = 0; bc.threads_per_cell = 0; bc.jit = 1;
    batch[d] = RS_BATCH_CREATE(&bc);
    if (!batch[d]) { fprintf(stderr, "rs_batch_create on device %d: %s\n", d, rs_last_error()); return 2; }
    /* global cell ids d*cells .. : the sharding rule of radiosaber_amd/sharding.py (seed.h commonSeed[0] = 805290992) */
    std::vector<uint32_t> seeds(cells);
    for (int c = 0; c < cells; ++c) {
      const uint64_t g = (uint64_t)d * cells + c;
      seeds[c] = (uint32_t)((g * 2654435761ull + 805290992ull) % 2147483647ull);
    }
    RS_OK_(rs_batch_seed(batch[d], seeds.data(), nullptr));
    RS_OK_(rs_batch_synthesize_cqi_at(batch[d], 0x5AB3, kHist, n_epochs, (int64_t)d * cells));
    RS_OK_(rs_batch_prepare_launch(batch[d], ttis)); /* the lean build of the kernel, outside the timed launches */
    HIP_OK(hipSetDevice(d));
    HIP_OK(hipMalloc((void**)&d_vec[d], sizeof(uint64_t) * S));
  }
  std::vector<ncclComm_t> comm(gpus);
  std::vector<int> devs(gpus);
  for (int d = 0; d < gpus; ++d) devs[d] = d;
  NCCL_OK(ncclCommInitAll(comm.data(), gpus, devs.data()));
  int rccl_version = 0;
  NCCL_OK(ncclGetVersion(&rccl_version));

  /* warm-up launch (code object load, first-touch), then the timed launches: every GPU's stream gets all of them at once */
  for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_run_async(batch[d], kWarm));
  for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_sync(batch[d]));
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < launches; ++k)
    for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_run_async(batch[d], ttis));
  for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_sync(batch[d]));
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  /* final aggregation: per-slice cumulative bytes, reduced per device, then one integer all-reduce over RCCL */
  for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_slice_bytes_device(batch[d], d_vec[d]));
  NCCL_OK(ncclGroupStart());
  for (int d = 0; d < gpus; ++d)
    NCCL_OK(ncclAllReduce(d_vec[d], d_vec[d], (size_t)S, ncclUint64, ncclSum, comm[d], (hipStream_t)rs_batch_stream(batch[d])));
  NCCL_OK(ncclGroupEnd());
  for (int d = 0; d < gpus; ++d) RS_OK_(rs_batch_sync(batch[d]));
  std::vector<uint64_t> total(S);
  HIP_OK(hipSetDevice(0));
  HIP_OK(hipMemcpy(total.data(), d_vec[0], sizeof(uint64_t) * S, hipMemcpyDeviceToHost));

  int bad = 0;
  if (check) {
    std::vector<uint64_t> want(S, 0), one(S), got(S);
    for (int d = 0; d < gpus; ++d) {
      RS_OK_(rs_batch_slice_bytes(batch[d], one.data()));
      for (int s = 0; s < S; ++s) want[s] += one[s];
    }
    for (int d = 0; d < gpus; ++d) {
      HIP_OK(hipSetDevice(d));
      HIP_OK(hipMemcpy(got.data(), d_vec[d], sizeof(uint64_t) * S, hipMemcpyDeviceToHost));
      for (int s = 0; s < S; ++s) bad += got[s] != want[s];
    }
    if (bad) fprintf(stderr, "CHECK FAILED: %d per-slice sums differ between the RCCL all-reduce and the host-side sum\n", bad);
  }
  const double sim_seconds = ((double)launches * ttis + kWarm) / 1000.0; /* one TTI = 1 ms; the counters include the warm-up epoch */
  const double n_cells_total = (double)gpus * cells;
  std::string mbps = "[", bytes = "[";
  for (int s = 0; s < S; ++s) {
    char buf[64];
    /* plot_throughput.py: bytes / seconds * 8 / 1e6, here averaged per cell */
    snprintf(buf, sizeof buf, "%s%.6f", s ? ", " : "", (double)total[s] * 8.0 / 1e6 / sim_seconds / n_cells_total);
    mbps += buf;
    snprintf(buf, sizeof buf, "%s%llu", s ? ", " : "", (unsigned long long)total[s]);
    bytes += buf;
  }
  mbps += "]"; bytes += "]";
  printf("{\"program\": \"rs_multi_gpu\", \"n_gpus\": %d, \"cells_per_gpu\": %d, \"ttis_per_launch\": %d, \"launches\": %d, "
         "\"sched\": %d, \"slices\": %d, \"ues\": %d, \"rbgs\": %d, \"wall_s\": %.6f, \"value\": %.1f, \"unit\": \"TTIs/s\", "
         "\"rccl_version\": %d, \"reduced_with\": \"ncclAllReduce(ncclUint64, ncclSum) over ncclCommInitAll(%d)\", "
         "\"check\": %s, \"slice_bytes\": %s, \"slice_mbps_per_cell\": %s}\n",
         gpus, cells, ttis, launches, sched, S, U, R, wall, n_cells_total * launches * ttis / wall, rccl_version, gpus,
         check ? (bad ? "\"FAILED\"" : "\"ok\"") : "null", bytes.c_str(), mbps.c_str());
  for (int d = 0; d < gpus; ++d) {
    (void)ncclCommDestroy(comm[d]);
    (void)hipSetDevice(d);
    (void)hipFree(d_vec[d]);
    rs_batch_destroy(batch[d]);
  }
  return bad ? 1 : 0;
}
