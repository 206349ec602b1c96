#!/usr/bin/env python3
"""bench.py -- scheduled TTIs/s of the RadioSaber downlink RBG allocation path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own N ranks: the parent process touches
neither HIP nor torch, runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port <free> bench.py ...` as a CHILD (no exec), relays rank 0's single JSON line and exits with the child's code.

Workload (BASELINE.json configs[3]/[4], the batched form of the metric's 20 x 500 x 25 grid):
every GPU holds `--cells` (512) independent cells of 20 slices x 25 UEs (500 UEs) x 25 RBGs,
scheduler 9 (RadioSaber / MaximizeCell), PF parameters epsilon=1 psi=1, weights 0.05, backlogged
flows, synthetic per-RBG CQI drawn i.i.d. from the reference trace corpus' histogram and redrawn
every 40 TTIs, one libc-compatible rand() stream per cell.  One STEP = one kernel launch that runs
`--ttis` (8000) complete DoSchedule() iterations of every cell; all inputs (the CQI grids of every
epoch, the cell state) are resident in HBM before the timed region starts.  The default timed region
is 20 launches = 160 000 TTIs per cell, about four seconds.

value = cells_total * ttis * steps / wall time, wall time bracketed by barrier + device sync on
both sides, max over ranks.  Cells are independent, so ranks share nothing during the run (cell ids,
rand() seeds and CQI grids are functions of the GLOBAL cell id); the only collective is the final
all-reduce (RCCL) of the per-slice cumulative byte counters uint64[S].

The JSON line also carries
  roofline        SURVEY.md 8d as the contract defines it: algorithmic bytes (B_TTI = U*R + 16U + 4U + 4R + 16S per
                  cell-TTI, "streamed-CQI" accounting) of one launch / that launch's HIP-event duration, against the
                  8 TB/s HBM3E peak -- a NOMINAL figure: the kernel keeps the CQI grid in LDS for 40 TTIs, so
                  `resident_bytes_per_cell_tti` (compulsory traffic of the resident design) and `traffic` (HBM bytes per
                  launch from the PMC passes recorded in profiles/traffic.json, NOT measured in this run: see
                  `traffic_source`) say what actually moves;
  roofline_issue  what really bounds the kernel: wave-instructions per cell-TTI (SQ_INSTS_* from profiles/inst_counts.json,
                  tools/pmc_insts.sh) x this run's TTIs/s against the chip's VALU issue rate;
  roofline.streamed   the same batch with `cqi_refresh = 1` -- a new CQI grid from HBM every TTI, SURVEY 8d's "streamed-CQI" mode:
                  the one figure of the line where moved bytes ~ algorithmic bytes, i.e. real bytes / real time (N = 1 only;
                  a bounded set of epochs cycles, larger than the 256 MiB Infinity Cache; `--cqi-refresh R` runs the MAIN batch
                  in that mode, which is how tools/profile_streamed.sh takes its PMC passes);
  value_r64       the same batch on the as-shipped 64-RBG grid (N = 1 only);
  value_cells1024 the same workload with 1 024 cells (all the wave slots the LDS carve allows), `occupancy` says what the headline leaves empty;
  cpu_baseline    the CPU oracle (oracle/, the bit-exact restatement of the reference) timed on this box's host cores with
                  OpenMP over independent cells (rank 0, N = 1 only);
  parity_sample   the first 32 global cells of the workload x 2 000 TTIs run again on the GPU (untimed, same kernel build) and on the
                  oracle: {"cells": 32, "ttis": 2000, "bit_exact": true}; the run exits with code 3 when they differ.
"""
import os

# the host driver only supports dmabuf IPC: must be in the environment before HIP/HSA initialises (RCCL across processes)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402
from pathlib import Path  # noqa: E402

import numpy as np  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue peak: 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction (same guide, cycle-constants table)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2


def algorithmic_bytes_per_tti(U, R, S, sched=9):
    """SURVEY.md 8(d), per scheduler: the compulsory bytes of ONE TTI of one cell.

    The transport schedulers (8, 9, 10, 101, 103; downlink-transport-scheduler.cpp:530-567) rank EVERY user on every RBG:
        U*R (cqi u8 read) + 8U + 8U (avg_rate read / write) + 4U (bytes out) + 4R (rbg -> user out) + 2*S*8 (slice_rbs_offset_ in / out)
    Per-flow PF (1; downlink-packet-scheduler.cpp:179-331) does the same without any slice state: the 16 S bytes go.
    NVS (7, 11; downlink-nvs-scheduler.cpp:144-194 SelectFlowsToSchedule, :275-358) serves ONE slice per TTI: only that slice's rows of
    the grid and that slice's averages take part (U/S users on average); what it keeps per slice is slice_ewma_time_ (in / out).
    Round 5 applied the first formula to every scheduler, which printed nominal fractions above 1 for scheduler 7."""
    if sched in (7, 11):
        n = U / S  # users of the served slice (the mean over slices for a ragged configuration)
        return n * R + 8 * n + 8 * n + 4 * n + 4 * R + 2 * S * 8
    if sched == 1:
        return U * R + 8 * U + 8 * U + 4 * U + 4 * R
    return U * R + 8 * U + 8 * U + 4 * U + 4 * R + 2 * S * 8


def resident_bytes_per_tti(U, R, S, n_ttis, refresh=40):
    """Compulsory HBM traffic of the resident design per cell-TTI: the CQI grid once per refresh interval, the cell
    state (avg f64, tx i32 in and out; cumulative bytes / RBs i64 read-modify-write; slice state; scalars) once per launch."""
    state = U * (8 + 4) * 2 + U * 16 * 2 + S * 16 + 2 * 200
    return U * R / refresh + state / n_ttis


def _physical_cores():
    """Distinct (package, core) pairs this process may run on."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    seen = set()
    for cpu in allowed:
        base = Path(f"/sys/devices/system/cpu/cpu{cpu}/topology")
        try:
            seen.add(((base / "physical_package_id").read_text().strip(), (base / "core_id").read_text().strip()))
        except OSError:
            seen.add(("?", str(cpu)))
    return max(1, len(seen)), len(allowed)


def _cgroup_cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota); None = unlimited."""
    try:
        q, per = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(args, slices, seeds):
    """The oracle on the host cores: independent cells spread with OpenMP inside librs_oracle.so (rso_run_synth_many), one
    thread per physical core; single-thread rate beside it.  BASELINE.md 3 / SURVEY 8(d)."""
    from oracle import oracle_py as O
    O.lib()
    hist = np.asarray(args.hist, np.float64)
    p = hist / hist.sum()
    U, R = slices.n_users, args.rbgs
    rng = np.random.default_rng(1000)

    def grids_for(n_ttis):
        return rng.choice(np.arange(1, 16, dtype=np.uint8), size=((n_ttis + 39) // 40, U, R), p=p).astype(np.uint8)

    tmpl = O.Cell(slices.ues_per_slice, R, args.rbg_size, args.sched, weights=slices.weight)
    # one thread, one cell: the single-core rate
    g = grids_for(400)
    t0 = time.perf_counter()
    O.run_synth_many(tmpl, 1, g, seeds[:1], 400, threads=1)
    per_core = 400 / (time.perf_counter() - t0)
    physical, logical = _physical_cores()
    quota = _cgroup_cpu_quota()
    # one thread per physical core the container may actually use (a cgroup CPU quota caps the usable cores below nproc)
    cores = physical if quota is None else max(1, min(physical, int(quota + 0.5)))
    n_cells = 2 * cores

    def run(n_ttis):
        sd = np.resize(np.asarray(seeds, np.uint32), n_cells)
        gg = grids_for(n_ttis)
        t0 = time.perf_counter()
        _, used = O.run_synth_many(tmpl, n_cells, gg, sd, n_ttis, threads=cores)
        return n_cells * n_ttis / (time.perf_counter() - t0), used

    probe_rate, _ = run(200)  # short all-core probe: sizes the timed sample to ~15 s (--cpu-baseline-seconds) whatever the box delivers
    n_ttis = int(max(200, min(40000, args.cpu_baseline_seconds * probe_rate / n_cells)))
    rate, used = run(n_ttis)
    wall = n_cells * n_ttis / rate
    eff = rate / (per_core * used)
    out = {"value": rate, "unit": "TTIs/s", "cores": used, "kind": "port",
           "single_core_value": per_core, "scaling_vs_linear": eff, "cpu_model": _cpu_model(),
           "physical_cores": physical, "logical_cpus": logical, "cgroup_cpu_quota": quota,
           "sample": f"{n_cells} independent cells x {n_ttis} TTIs of the same workload, OpenMP over cells inside the "
                     f"oracle (schedule(dynamic,1)), {used} threads = one per usable physical core "
                     f"({physical} physical cores, cgroup CPU quota {quota}); wall {wall:.1f} s"}
    if eff < 0.5:
        out["warning"] = (f"all-core rate is only {eff:.2f} x linear ({used} threads x {per_core:.0f} TTIs/s single-core): "
                          "the baseline is not core-bound on this box")
        print("cpu_baseline WARNING: " + out["warning"], file=sys.stderr)
    return out


def parity_sample(rs, args, slices, sharding, device, n_cells=32, n_ttis=2000):
    """The line proves its own run: the first `n_cells` GLOBAL cells of the workload -- same seeds, same device-synthesised
    CQI grids, same shape-specialised kernel (a launch this long runs its lean build, like the timed ones) -- for `n_ttis` TTIs on the GPU,
    outside the timed region, against the oracle on the host cores: PF averages bit for bit, cumulative bytes / RBs, slice state.
    The oracle is the checker here, never the thing measured or shipped."""
    from oracle import oracle_py as O
    O.lib()
    n_epochs = (n_ttis + args.cqi_refresh - 1) // args.cqi_refresh
    b = rs.BatchScheduler(slices, args.rbgs, args.rbg_size, n_cells, sched=args.sched, device=device, threads_per_cell=args.threads,
                          jit=not args.no_jit, cqi_refresh=args.cqi_refresh)
    ids = sharding.cell_ids_for_rank(0, 1, n_cells)
    seeds = sharding.seeds_for_cells(ids)
    b.seed(seeds)
    b.synthesize_cqi(0x5AB3, n_epochs, first_cell=0)
    grids = np.stack([b.download_cqi_epochs(c) for c in range(n_cells)])
    b.prepare_launch(n_ttis)
    b.run(n_ttis)
    st = b.state()
    kernel, status = b.kernel_name, b.jit_status()
    b.close()
    tmpl = O.Cell(slices.ues_per_slice, args.rbgs, args.rbg_size, args.sched, weights=slices.weight, epsilon=slices.algo_epsilon, psi=slices.algo_psi)
    t0 = time.perf_counter()
    ref = O.run_synth_cells(tmpl, grids, seeds, n_ttis, refresh=args.cqi_refresh)
    oracle_s = time.perf_counter() - t0
    differ = [k for k in ("cum_bytes", "cum_rbs") if not np.array_equal(st[k], ref[k])]
    differ += [k for k in ("avg_rate", "slice_state") if st[k].tobytes() != ref[k].tobytes()]
    return {"cells": n_cells, "ttis": n_ttis, "bit_exact": not differ, "differs_in": differ, "compared": "PF averages (bitwise), cumulative bytes, "
            "cumulative RBs, slice state (bitwise) of every cell after the run", "global_cell_ids": [int(ids[0]), int(ids[-1])],
            "kernel": kernel, "jit_status": list(status), "oracle_seconds": round(oracle_s, 2), "oracle_threads": ref["threads"],
            "total_bytes": int(st["cum_bytes"].sum())}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _profiler_in_environment():
    """rocprofv3 (or another rocprofiler-sdk tool) is preloaded into this process: its library has initialised the GPU here."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprofiler" in pre or "rocprof" in pre:
        return "LD_PRELOAD=" + pre
    for k in os.environ:
        if k.startswith(("ROCPROF", "ROCPROFILER", "ROCP_")):
            return k
    return None


def _launch_own_ranks(n):
    """`python bench.py --gpus N` (N > 1) without a launcher: N fresh rank processes under torch.distributed.run, started as a
    CHILD of this process (never exec: a process that replaces itself after touching the GPU takes the box down; this parent
    has not touched HIP, and still does not exec).  stdout of the children is relayed: the one JSON line of rank 0 goes to
    stdout, anything else to stderr.  Returns the children's exit code."""
    import subprocess
    prof = _profiler_in_environment()
    if prof:
        # the profiler's preloaded library has already initialised the GPU in THIS process: starting the ranks from here is
        # the hop this pool forbids.  Profile one rank, or start the ranks under the launcher and profile inside it.
        print(f"bench.py: refusing to start {n} ranks from a process a profiler is loaded into ({prof}): profile a single "
              "rank (`rocprofv3 ... -- python3 bench.py`), or run `python -m torch.distributed.run ... bench.py --gpus N` "
              "yourself", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")  # torch.distributed.run would set 1 and warn; the ranks' host work is tiny
    print("bench.py: starting " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    n_lines = 0
    for ln in proc.stdout:
        is_line = False
        if ln.lstrip().startswith("{"):
            try:
                is_line = "metric" in json.loads(ln)
            except ValueError:
                is_line = False
        if is_line:
            n_lines += 1
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and n_lines != 1:
        print(f"bench.py: expected one JSON line from rank 0, saw {n_lines}", file=sys.stderr)
        return 1
    return rc


def _rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:  # reporting nicety only
        return f"unknown ({e})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help="timed launches; the default measures 160 000 TTIs per cell, about 4 s")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=512, help="independent cells per GPU")
    ap.add_argument("--ttis", type=int, default=8000,
                    help="TTIs per step (per launch; at most 32 768): 8 000 makes one step ~0.2 s, so that even a "
                         "20-step run is seconds long and visible to an external GPU-activity sampler")
    ap.add_argument("--slices", type=int, default=20)
    ap.add_argument("--ues-per-slice", type=int, default=25)
    ap.add_argument("--rbgs", type=int, default=25)
    ap.add_argument("--rbg-size", type=int, default=4)
    ap.add_argument("--sched", type=int, default=9)
    ap.add_argument("--threads", type=int, default=0, help="workgroup size per cell (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs: cpu_baseline and parity_sample")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0, help="CPU work of the timed cpu_baseline sample (default ~15 s)")
    ap.add_argument("--no-r64", action="store_true", help="skip the extra 64-RBG measurement (value_r64)")
    ap.add_argument("--no-cells1024", action="store_true", help="skip the extra 1 024-cell measurement (value_cells1024)")
    ap.add_argument("--cqi-refresh", type=int, default=40,
                    help="TTIs between two CQI grids (reference: 40, enb-mac-entity.cc:38).  1 = streamed-CQI mode: a new grid "
                         "from HBM every TTI; the epochs then cycle through a bounded set (rs_batch_config.cqi_epoch_wrap)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the extra streamed-CQI measurement (roofline.streamed)")
    ap.add_argument("--no-jit", action="store_true", help="use the kernels built into the library instead of the "
                    "shape-specialised one compiled at create time")
    ap.add_argument("--config-key", default=None,
                    help="one of the reference's shipped experiment configurations (a key of tests/golden/experiment_configs.json, "
                         "e.g. exp-fixranues/20slices/config-pf.json): ragged ues_per_slice, weights and algo parameters from the "
                         "fixture, on the as-shipped 64-RBG grid unless --rbgs says otherwise; every flow backlogged")
    ap.add_argument("--allow-variant", action="store_true",
                    help="run although RS_JIT_EXTRA / RS_JIT is set in the environment (tuning experiments; the line says so)")
    args = ap.parse_args()

    # A headline must come from the product kernel: build switches in the environment are refused unless asked for, and the
    # JSON line always says what the run-time compiler was given.
    jit_extra = os.environ.get("RS_JIT_EXTRA", "")
    jit_env = os.environ.get("RS_JIT")
    if (jit_extra.strip() or jit_env is not None) and not args.allow_variant:
        raise SystemExit(f"bench.py: RS_JIT_EXTRA={jit_extra!r} RS_JIT={jit_env!r} is set: the kernel would not be the product "
                         "build; unset it, or pass --allow-variant for a tuning experiment (recorded as jit_extra in the line)")

    # a launcher (torch.distributed.run) exports WORLD_SIZE, RANK and LOCAL_RANK together; a stray WORLD_SIZE alone is not one
    under_launcher = all(k in os.environ for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"))
    world = int(os.environ["WORLD_SIZE"]) if under_launcher else 1
    if not under_launcher and args.gpus > 1:
        # no launcher around us: start the ranks ourselves, BEFORE anything in this process touches HIP (or imports torch)
        raise SystemExit(_launch_own_ranks(args.gpus))

    # The run-time kernels of this process are built by the toolchain the library itself was built with (radiosaber_amd/toolchain.py:
    # the system's comgr is mapped before torch brings its wheel's copy; RS_SYSTEM_COMGR=0 leaves it to the import order).  The line's
    # `compiler` field says which compiler it was.
    from radiosaber_amd import toolchain
    comgr_in_use = toolchain.prefer_system_compiler()
    import torch
    import torch.distributed as dist

    import radiosaber_amd as rs
    from radiosaber_amd import sharding

    args.hist = rs.TRACE_CQI_HISTOGRAM
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # launched under torch.distributed.run: its world size is the number of GPUs
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # one rank per GPU; RS_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a box with fewer GPUs than ranks
    backend = os.environ.get("RS_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPU(s) visible: RCCL needs one GPU per rank "
                         "(RS_BENCH_BACKEND=gloo runs the N > 1 code path with ranks sharing a GPU, for plumbing tests only)")
    local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    ranks_in_group = 1
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        ranks_in_group = dist.get_world_size()

    red_dev = "cuda" if backend == "nccl" else "cpu"  # where the tiny reductions live
    if args.config_key:
        cfgs = json.loads((ROOT / "tests" / "golden" / "experiment_configs.json").read_text())
        if args.config_key not in cfgs:
            raise SystemExit(f"bench.py: --config-key {args.config_key!r} is not in tests/golden/experiment_configs.json ({len(cfgs)} keys, "
                             f"e.g. {sorted(cfgs)[0]})")
        c = cfgs[args.config_key]
        slices = rs.SliceConfig(c["ues_per_slice"], weight=c["weight"], algo_epsilon=c["algo_epsilon"], algo_psi=c["algo_psi"])
        if not any(a.startswith("--rbgs") for a in sys.argv):
            args.rbgs, args.rbg_size = 64, 8  # the reference's 100 MHz carrier: 512 PRBs in RBGs of 8
        args.slices, args.ues_per_slice = slices.n_slices, None
        S, U, R = slices.n_slices, slices.n_users, args.rbgs
    else:
        S, U, R = args.slices, args.slices * args.ues_per_slice, args.rbgs
        slices = rs.SliceConfig([args.ues_per_slice] * S, weight=[1.0 / S] * S)
    want_jit = not args.no_jit

    EPOCH_BYTES_CAP = 8 << 30  # HBM spent on CQI epochs before they start to cycle (far beyond the 256 MiB Infinity Cache)

    def make_batch(n_rbgs, rbg_size, launches, ttis, refresh=40):
        need = (launches * ttis + refresh - 1) // refresh
        k8 = (U + 7) // 8
        stride = (8 * (k8 if k8 & 1 else k8 + 1) * n_rbgs + 15) // 16 * 16  # device-resident grids are RBG-major [R][Upad]
        n_epochs = need
        wrap = False
        if need * args.cells * stride > EPOCH_BYTES_CAP:  # (at the reference's 40-TTI refresh too: --steps is not bounded by HBM)
            n_epochs, wrap = max(2, EPOCH_BYTES_CAP // (args.cells * stride)), True
        b = rs.BatchScheduler(slices, n_rbgs, rbg_size, args.cells, sched=args.sched, device=local_rank,
                              threads_per_cell=args.threads, jit=want_jit, cqi_refresh=refresh, cqi_epoch_wrap=wrap)
        b.n_epochs_resident, b.epoch_stride, b.epochs_wrap = n_epochs, stride, wrap
        # cell ids are global: rank r owns cells [r*cells, (r+1)*cells); seeds and CQI grids are keyed on them
        b.seed(sharding.seeds_for_cells(sharding.cell_ids_for_rank(rank, world, args.cells)))
        b.synthesize_cqi(0x5AB3, n_epochs,  # generated on the device, stay in HBM
                         first_cell=sharding.first_cell_for_rank(rank, world, args.cells))
        b.prepare_launch(ttis)  # the lean build of the kernel is compiled -- and both builds self-checked against the built-in kernels -- here, not inside a (possibly timed) first launch
        # (only now: a build that the self-check rejects is dropped in prepare_launch, and the batch would run on the built-in kernels)
        code, msg = b.jit_status()
        ok = (not want_jit) or code == 1
        if world > 1:
            # every rank leaves together: one rank exiting alone would leave the others in the next collective forever
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            all_ok = bool(flag.item())
        else:
            all_ok = ok
        if not all_ok:
            # a headline measured on the slower built-in kernel must not pass unnoticed
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(f"bench.py: the shape-specialised kernel was requested but is not in use on "
                             f"{'this rank' if not ok else 'another rank'} ({msg or code}); "
                             "rerun with --no-jit to measure the built-in kernels on purpose")
        return b

    batch = make_batch(R, args.rbg_size, args.steps + args.warmup, args.ttis, args.cqi_refresh)
    seeds = sharding.seeds_for_cells(sharding.cell_ids_for_rank(rank, world, args.cells))

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.run(args.ttis)
    sync_all()
    t0 = time.perf_counter()
    ms = batch.run_timed(args.ttis, args.steps)  # K launches, HIP events on the launch stream
    sync_all()
    wall = time.perf_counter() - t0
    rank_kernel_ms = [float(np.mean(ms))]
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
        # every rank's mean launch duration (HIP events), so that a straggler GPU shows in the line
        km = torch.zeros(world, dtype=torch.float64, device=red_dev)
        km[rank] = float(np.mean(ms))
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        rank_kernel_ms = [round(float(x), 4) for x in km.tolist()]

    # final aggregation: per-slice cumulative bytes, reduced over the GPUs (RCCL over xGMI)
    slice_bytes = torch.zeros(S, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    batch.slice_bytes_into(slice_bytes.data_ptr())
    batch.sync()
    if red_dev == "cpu":
        slice_bytes = slice_bytes.cpu()
    sharding.all_reduce_slice_bytes(slice_bytes, dist if world > 1 else None)
    total_bytes = int(slice_bytes.sum().item())
    kernel_name = batch.kernel_name
    batch_epochs, batch_wrap = batch.n_epochs_resident, batch.epochs_wrap
    jit_code, jit_msg = batch.jit_status()  # (before close: a closed batch has no status to read)
    clk_mhz, cell_ms = batch.debug_clocks()  # the last timed launch, from the kernel's own s_memtime / s_memrealtime readings
    batch.close()

    parity_failed = False
    if rank == 0:
        total_ttis = world * args.cells * args.ttis * args.steps
        value = total_ttis / wall
        launch_s = float(np.mean(ms)) / 1e3
        b_tti = algorithmic_bytes_per_tti(U, R, S, args.sched)
        achieved = b_tti * args.cells * args.ttis / launch_s / 1e9
        main_key = key = f"sched{args.sched}_S{S}_U{U}_R{R}_cells{args.cells}" + ("" if args.cqi_refresh == 40 else f"_refresh{args.cqi_refresh}")
        src_hash = rs.device_source_hash()  # identity of the kernel sources inside the library that just ran

        def recorded(fname, key=None):
            """Entry of profiles/<fname> for this workload + whether it was recorded on other kernel code than this run's."""
            key = key or main_key
            f = ROOT / "profiles" / fname
            ent = json.loads(f.read_text()).get(key) if f.exists() else None
            if not ent:
                return None, None
            stale = ent.get("source_hash") != src_hash
            if stale:
                print(f"bench.py: profiles/{fname}[{key}] was recorded on device sources {ent.get('source_hash', '(no hash)')}, "
                      f"this library is {src_hash}: STALE (re-run tools/profile_round.sh / tools/pmc_insts.sh)", file=sys.stderr)
            return ent, stale

        traffic = traffic_src = traffic_stale = None
        ent, traffic_stale = recorded("traffic.json")
        if ent:  # recorded per TTI per cell so that it scales to this run's launch length
            traffic = ent["hbm_bytes_per_cell_tti"] * args.cells * args.ttis
            traffic_src = (f"profiles/traffic.json[{key}]@{ent.get('commit', '?')} source_hash {ent.get('source_hash', 'none')} "
                           "(PMC passes of an earlier run, not this one)")
        try:  # attainable HBM rate of this GPU (16 B/lane streaming copy, 1 GiB each way), beside the spec peak
            copy_gbs = rs.hbm_copy_probe(local_rank, 1 << 30, 10)
        except Exception as e:  # measurement nicety only
            print(f"hbm_copy_probe failed: {e}", file=sys.stderr)
            copy_gbs = None
        line = {
            "metric": "scheduled TTIs/sec (and us/TTI) at 20 slices x 500 UEs x 25 RBGs; HBM GB/s",
            "value": value, "unit": "TTIs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"sched={args.sched} ({'RadioSaber/MaximizeCell' if args.sched == 9 else 'see --sched'}), "
                                   f"{args.cells} independent cells per GPU x ({S} slices x "
                                   f"{args.ues_per_slice if args.ues_per_slice else 'the shipped ' + str(args.config_key) + ' mix of'} UEs "
                                   f"= {U} UEs x {R} RBGs), {args.ttis} TTIs per step, CQI i.i.d. from the trace "
                                   f"histogram redrawn every {args.cqi_refresh} TTIs (BASELINE.json configs[3]; configs[4] at 8 GPUs)",
                       "cqi_refresh": args.cqi_refresh, "cqi_epochs_resident": batch_epochs, "cqi_epochs_cycle": batch_wrap,
                       "config_key": args.config_key, "ues_per_slice": slices.ues_per_slice,
                       "cells_per_gpu": args.cells, "ttis_per_step": args.ttis, "slices": S, "ues": U, "rbgs": R,
                       "sched": args.sched, "parallelism": f"cells sharded over {world} GPU(s), no data-path collective"},
            "us_per_tti_per_cell": launch_s / args.ttis * 1e6,
            "kernel": kernel_name,
            # unlogged launches of >= RS_JIT_LEAN_MIN_TTIS (256) TTIs on epoch grids run the LEAN build of that kernel: the launch's unused
            # run-time options (trace rows, per-PRB twins, decision log, error-model draws) compiled out, results identical
            "kernel_build": ("lean" if (want_jit and os.environ.get("RS_JIT_LEAN", "1") != "0"
                                        and args.ttis >= int(os.environ.get("RS_JIT_LEAN_MIN_TTIS", "256"))
                                        and "lean build unavailable" not in jit_msg) else "general"),
            "jit_status": [jit_code, jit_msg],
            # who reduced: the process-group backend ("nccl" IS RCCL on ROCm), the ranks it saw, the library version
            "backend": backend if world > 1 else None, "ranks_in_group": ranks_in_group,
            "rccl_version": _rccl_version(torch) if (world > 1 and backend == "nccl") else None,
            "jit_extra": jit_extra, "jit_env": jit_env,
            "kernel_ms_per_launch": [round(float(x), 4) for x in ms],
            "kernel_ms_mean_per_rank": rank_kernel_ms,
            # the shader clock the last timed launch ran at (mean over the cells; the kernel reads both of its clocks at its first and
            # last instruction) and the longest single cell's own run time: a lease whose GPU clocks lower shows here, not as a mystery
            "shader_mhz": float(np.mean(clk_mhz)), "cell_ms_max": float(np.max(cell_ms)), "cell_ms_mean": float(np.mean(cell_ms)),
            "cell_ms_min": float(np.min(cell_ms)), "compute_units": int(torch.cuda.get_device_properties(local_rank).multi_processor_count),
            "total_slice_bytes": total_bytes,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale,
                         "traffic_frac_of_peak": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "algorithmic_bytes_per_cell_tti": b_tti,
                         "resident_bytes_per_cell_tti": resident_bytes_per_tti(U, R, S, args.ttis, args.cqi_refresh),
                         "note": "achieved/frac use SURVEY 8d's streamed-CQI bytes as the contract asks; the kernel keeps the "
                                 "grid in LDS between refreshes, so `traffic` (PMC) is what moves and the kernel is "
                                 "issue/latency-bound: see roofline_issue",
                         "measured_copy_gbs": copy_gbs},
        }
        line["source_hash"] = src_hash
        line["compiler"] = rs.jit_compiler_identity()  # the hiprtc / clang that built the timed kernel (part of every cache key)
        line["comgr"] = comgr_in_use
        ent, inst_stale = recorded("inst_counts.json")
        if ent:
            rate = value / world  # per GPU
            line["roofline_issue"] = {
                "stale": inst_stale,
                "bound": "valu-issue + dependent-chain latency",
                "valu_per_cell_tti": ent["valu"], "salu_per_cell_tti": ent["salu"], "lds_per_cell_tti": ent["lds"],
                "valu_issue_frac": ent["valu"] * rate / VALU_ISSUE_PEAK,
                "valu_issue_peak_per_s": VALU_ISSUE_PEAK,
                "phase_shares": ent.get("phase_shares"),
                "source": f"profiles/inst_counts.json[{key}]@{ent.get('commit', '?')} source_hash {ent.get('source_hash', 'none')} "
                          "(SQ_INSTS_* per cell-TTI from tools/pmc_insts.sh, phase shares from tools/phase_stamps.py; counts "
                          "are per build, the rate is this run's)"}
        if world == 1 and not args.no_streamed and args.cqi_refresh != 1:
            # SURVEY 8(d) asks for both modes: the same cells with a new CQI grid from HBM EVERY TTI (cqi_refresh = 1), where the
            # bytes the kernel moves are ~ the algorithmic bytes: the one roofline figure of this line that is real bytes / real time
            st_ttis = min(args.ttis, 2000)
            bs = make_batch(R, args.rbg_size, 4, st_ttis, refresh=1)
            bs.run(st_ttis)
            ms_s = bs.run_timed(st_ttis, 3)
            st_epochs, st_stride = bs.n_epochs_resident, bs.epoch_stride
            bs.close()
            st_value = args.cells * st_ttis / (float(np.mean(ms_s)) / 1e3)
            skey = f"sched{args.sched}_S{S}_U{U}_R{R}_cells{args.cells}_refresh1"
            ent, stale = recorded("traffic.json", skey)
            moved = ent["hbm_bytes_per_cell_tti"] if ent else None
            line["roofline"]["streamed"] = {
                "value": st_value, "unit": "TTIs/s", "us_per_tti_per_cell": float(np.mean(ms_s)) * 1e3 / st_ttis,
                "achieved_gbs": b_tti * st_value / 1e9, "frac": b_tti * st_value / 1e9 / HBM_PEAK_GBS,
                "traffic": moved * args.cells * st_ttis if moved else None, "traffic_bytes_per_cell_tti": moved,
                "traffic_gbs": moved * st_value / 1e9 if moved else None,
                "traffic_frac_of_peak": moved * st_value / 1e9 / HBM_PEAK_GBS if moved else None,
                "traffic_stale": stale,
                "traffic_source": f"profiles/traffic.json[{skey}]@{ent.get('commit', '?')}" if ent else None,
                "epochs_resident": st_epochs, "epoch_bytes_resident": st_epochs * args.cells * st_stride,
                "ttis_per_launch": st_ttis, "launches": 3,
                "note": "cqi_refresh = 1: every TTI loads its grid from HBM (epochs cycle through a set larger than the Infinity "
                        "Cache); achieved_gbs = algorithmic bytes x this rate, traffic_* = PMC bytes of the same mode"}
        if world == 1 and not args.no_r64 and (R, args.rbg_size) != (64, 8):
            # the as-shipped grid: 100 MHz = 512 PRBs = 64 RBGs of 8 (SURVEY 8d asks for it beside the headline)
            b64 = make_batch(64, 8, 4, args.ttis)
            b64.run(args.ttis)
            ms64 = b64.run_timed(args.ttis, 3)
            b64.close()
            line["value_r64"] = args.cells * args.ttis / (float(np.mean(ms64)) / 1e3)
            line["us_per_tti_per_cell_r64"] = float(np.mean(ms64)) * 1e3 / args.ttis
            # the same two records for that shape: nominal (algorithmic) rate and what the PMC passes saw move
            key = f"sched{args.sched}_S{S}_U{U}_R64_cells{args.cells}"
            b64_tti = algorithmic_bytes_per_tti(U, 64, S, args.sched)
            ach64 = b64_tti * line["value_r64"] / 1e9
            ent, stale = recorded("traffic.json", key)
            line["roofline_r64"] = {"bound": "hbm", "achieved": ach64, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach64 / HBM_PEAK_GBS,
                                    "algorithmic_bytes_per_cell_tti": b64_tti,
                                    "traffic_bytes_per_cell_tti": ent["hbm_bytes_per_cell_tti"] if ent else None,
                                    "traffic_stale": stale,
                                    "traffic_source": f"profiles/traffic.json[{key}]@{ent.get('commit', '?')}" if ent else None}
        if world == 1 and not args.no_cells1024 and args.cells != 1024:
            # The headline's 512 cells are two workgroups of 8 waves on every CU -- 16 of its 32 wave slots (BASELINE configs[3] fixes the
            # 512); LDS would hold four.  The same workload with 1 024 cells shows what the idle slots are worth, so that
            # roofline_issue.valu_issue_frac is not read as "the kernel leaves two thirds of the chip idle".
            keep = args.cells
            args.cells = 1024
            b1k = make_batch(R, args.rbg_size, 4, args.ttis, args.cqi_refresh)
            b1k.run(args.ttis)
            ms1k = b1k.run_timed(args.ttis, 3)
            b1k.close()
            args.cells = keep
            line["value_cells1024"] = 1024 * args.ttis / (float(np.mean(ms1k)) / 1e3)
        cus = line["compute_units"]
        lds = rs.lds_bytes_per_cell(S, U, R, args.sched, args.threads or 512)
        line["occupancy"] = {"cells_per_cu": args.cells / cus, "threads_per_cell": args.threads or 512,
                             "waves_per_cu": args.cells / cus * ((args.threads or 512) // 64), "wave_slots_per_cu": 32,
                             "lds_bytes_per_cell": lds, "cells_per_cu_lds_allows": (160 * 1024) // lds,
                             "note": "BASELINE configs[3] fixes 512 cells: half of the CU's wave slots hold a wave; value_cells1024 is the same "
                                     "workload with the slots the LDS carve allows filled (the library picks 256 threads per cell there)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, slices, seeds)
            try:
                line["parity_sample"] = parity_sample(rs, args, slices, sharding, local_rank)
            except Exception as e:  # an oracle that cannot run is not a parity failure -- but it is said, and it is not "true"
                line["parity_sample"] = {"cells": 0, "ttis": 0, "bit_exact": None, "error": repr(e)}
        print(json.dumps(line), flush=True)
        if line.get("parity_sample", {}).get("bit_exact") is False:
            print("bench.py: the GPU run and the oracle DIFFER on the parity sample (" + ", ".join(line["parity_sample"]["differs_in"]) +
                  "): the line above is not a valid measurement", file=sys.stderr)
            parity_failed = True
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
