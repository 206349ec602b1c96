"""ctypes binding of include/radiosaber_hip.h.

The classes keep the reference's vocabulary (slices, UEs, RBGs, TTIs):

  SliceConfig      what the reference scheduler constructors read from the JSON config
                   (downlink-transport-scheduler.cpp:55-97): ues_per_slice + per-slice
                   weight / algo_alpha / algo_beta / algo_epsilon / algo_psi
  TtiScheduler     drop-in mode, one RBsAllocation() per call            (rs_create / rs_schedule_tti)
  BatchScheduler   many device-resident cells, whole DoSchedule() loops  (rs_batch_*)
"""
import ctypes as C
import json
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

RS_SCHED_SUBOPT = 101
RS_SCHED_PF, RS_SCHED_NVS, RS_SCHED_SEQUENTIAL, RS_SCHED_MAXCELL, RS_SCHED_UPPERBOUND, RS_SCHED_NVS_NONGREEDY, RS_SCHED_VOGEL = \
    1, 7, 8, 9, 10, 11, 103

# CQI histogram (CQI 1..15) of the reference's whole cqi-traces-noise0 corpus (158 traces x 475 rows
# x 512 PRBs; SURVEY.md 8d, re-counted by tools/make_trace_fixture.py)
TRACE_CQI_HISTOGRAM = (152600, 56656, 270880, 2088792, 3509504, 1595568, 4145392, 5295816, 1903424,
                       6890232, 4770864, 2842552, 3579624, 96000, 1227696)

_PKG = Path(__file__).resolve().parent
_LIB_PATH = Path(os.environ.get("RS_HIP_LIB", _PKG / "libradiosaber_hip.so"))


class RadioSaberError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"radiosaber_hip error {code}: {msg}")
        self.code = code


class _Config(C.Structure):
    _fields_ = [("n_slices", C.c_int32), ("n_users", C.c_int32), ("n_rbgs", C.c_int32),
                ("rbg_size", C.c_int32), ("sched", C.c_int32), ("device", C.c_int32),
                ("slice_weight", C.POINTER(C.c_double)), ("algo_alpha", C.POINTER(C.c_int32)),
                ("algo_beta", C.POINTER(C.c_int32)), ("algo_epsilon", C.POINTER(C.c_int32)),
                ("algo_psi", C.POINTER(C.c_int32)), ("user_to_slice", C.POINTER(C.c_int32)),
                ("stream", C.c_void_p), ("synthetic_exp", C.c_int32), ("link_tables", C.c_int32)]


class _BatchConfig(C.Structure):
    _fields_ = [("cell", _Config), ("n_cells", C.c_int32), ("first_tti", C.c_int32),
                ("cqi_refresh", C.c_int32), ("phy_error_draws", C.c_int32),
                ("threads_per_cell", C.c_int32), ("jit", C.c_int32),
                ("cqi_epoch_wrap", C.c_int32), ("queue_state_lds", C.c_int32), ("autotune", C.c_int32), ("selfcheck", C.c_int32)]


RS_LINK_DEFAULT, RS_LINK_HOST_LIBM, RS_LINK_PINNED_GLIBC_2_35 = 0, 1, 2  # rs_config.link_tables

RS_ABI_VERSION = 11  # the include/radiosaber_hip.h these ctypes structs mirror; passed to the *_checked create functions


class _TtiIn(C.Structure):
    _fields_ = [("n_users", C.c_int32), ("user_id", C.POINTER(C.c_int32)),
                ("cqi", C.POINTER(C.c_uint8)), ("avg_rate", C.POINTER(C.c_double)),
                ("rand0", C.c_int32), ("rand1", C.c_int32), ("cqi_prb", C.POINTER(C.c_uint8)),
                ("hol_delay", C.POINTER(C.c_double)), ("prio_has_data", C.POINTER(C.c_uint8)),
                ("rand_draws", C.POINTER(C.c_int32)),
                ("required_rbs", C.POINTER(C.c_int32)), ("data_to_transmit", C.POINTER(C.c_int32)), ("cqi_epoch", C.c_uint64)]


class _TtiOut(C.Structure):
    _fields_ = [("target_rbs", C.POINTER(C.c_int32)), ("quota_rbgs", C.POINTER(C.c_int32)),
                ("rbg_to_user", C.POINTER(C.c_int32)), ("user_nprb", C.POINTER(C.c_int32)),
                ("user_final_cqi", C.POINTER(C.c_int32)), ("user_mcs", C.POINTER(C.c_int32)),
                ("user_tbs_bits", C.POINTER(C.c_int32)),
                ("upper_rbg", C.POINTER(C.c_int32)), ("upper_user", C.POINTER(C.c_int32))]


class _BatchLog(C.Structure):
    _fields_ = [("rbg_to_user", C.POINTER(C.c_int16)), ("tbs_bits", C.POINTER(C.c_int32)),
                ("quota", C.POINTER(C.c_int16)), ("target", C.POINTER(C.c_int16)), ("uinfo", C.POINTER(C.c_int32)),
                ("slice_keys", C.POINTER(C.c_uint32))]


# every symbol include/radiosaber_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "rs_last_error", "rs_abi_version", "rs_device_count", "rs_link_tables",
    "rs_create", "rs_destroy", "rs_schedule_tti", "rs_get_slice_offset", "rs_set_slice_offset",
    "rs_batch_create", "rs_batch_destroy", "rs_batch_seed", "rs_batch_upload_cqi_epochs",
    "rs_batch_synthesize_cqi", "rs_batch_download_cqi_epochs", "rs_batch_set_trace",
    "rs_batch_run", "rs_batch_run_async", "rs_batch_sync", "rs_batch_run_logged",
    "rs_batch_run_timed", "rs_batch_read_state", "rs_batch_slice_bytes_device",
    "rs_batch_slice_bytes", "rs_jit_selfcheck", "rs_jit_selfcheck_queue", "rs_batch_debug_stamps", "rs_batch_ttis_done", "rs_batch_stream", "rs_batch_kernel_name",
    "rs_trace_read_mapping", "rs_trace_read_ue_log", "rs_trace_load_dir", "rs_hbm_copy_probe", "rs_lds_bytes_per_cell",
    "rs_get_rbg_size", "rs_dl_prbs_for_bandwidth", "rs_batch_synthesize_cqi_at", "rs_batch_run_logged_ex",
    "rs_batch_read_clock", "rs_batch_jit_status", "rs_batch_prepare_launch",
    "rs_batch_upload_cqi_epochs_prb", "rs_batch_set_trace_prb",
    "rs_batch_set_bearers", "rs_batch_set_arrivals", "rs_batch_read_bearer_state", "rs_internet_flow_arrivals",
    "rs_device_source_hash",
    "rs_create_checked", "rs_batch_create_checked", "rs_jit_selfcheck_untuned", "rs_batch_write_state",
    "rs_ctx_specialize", "rs_jit_selfcheck_dropin",
    "rs_batch_debug_heap_sorts", "rs_ctx_debug_heap_sorts",
    "rs_jit_cache_stats", "rs_jit_cache_file", "rs_jit_cache_warm", "rs_batch_autotune_report", "rs_batch_debug_clocks",
    "rs_batch_checkpoint_bytes", "rs_batch_checkpoint_save", "rs_batch_checkpoint_load",
    "rs_link_tables_pinned", "rs_link_tables_compare", "rs_ctx_jit_status", "rs_jit_compiler_identity",
]

_lib = None


def lib():
    """Load libradiosaber_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RadioSaberError(-100, f"{_LIB_PATH} is missing: run `python -m radiosaber_amd.build` "
                                    "(there is no CPU fallback)")
    L = C.CDLL(str(_LIB_PATH))
    L.rs_last_error.restype = C.c_char_p
    L.rs_link_tables.argtypes = [C.POINTER(C.c_double)] * 4
    L.rs_link_tables_pinned.argtypes = [C.POINTER(C.c_double)] * 4
    L.rs_link_tables_compare.argtypes = [C.c_char_p, C.c_size_t]
    L.rs_ctx_jit_status.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.rs_jit_compiler_identity.restype = C.c_char_p
    L.rs_jit_compiler_identity.argtypes = []
    L.rs_create.restype = C.c_void_p
    L.rs_create.argtypes = [C.POINTER(_Config)]
    L.rs_destroy.argtypes = [C.c_void_p]
    L.rs_schedule_tti.argtypes = [C.c_void_p, C.POINTER(_TtiIn), C.POINTER(_TtiOut)]
    L.rs_get_slice_offset.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.rs_set_slice_offset.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.rs_batch_create.restype = C.c_void_p
    L.rs_batch_create.argtypes = [C.POINTER(_BatchConfig)]
    L.rs_create_checked.restype = C.c_void_p
    L.rs_create_checked.argtypes = [C.POINTER(_Config), C.c_int, C.c_size_t]
    L.rs_batch_create_checked.restype = C.c_void_p
    L.rs_batch_create_checked.argtypes = [C.POINTER(_BatchConfig), C.c_int, C.c_size_t]
    L.rs_jit_selfcheck_untuned.argtypes = [C.c_int] * 6 + [C.c_char_p, C.c_size_t]
    L.rs_jit_selfcheck_dropin.argtypes = [C.c_int] * 6 + [C.c_char_p, C.c_size_t]
    L.rs_ctx_specialize.argtypes = [C.c_void_p]
    L.rs_batch_destroy.argtypes = [C.c_void_p]
    L.rs_batch_seed.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int64)]
    L.rs_batch_upload_cqi_epochs.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32]
    L.rs_batch_synthesize_cqi.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.c_int32]
    L.rs_batch_synthesize_cqi_at.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_double), C.c_int32, C.c_int64]
    L.rs_batch_run_logged_ex.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_BatchLog)]
    L.rs_batch_read_clock.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rs_batch_jit_status.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.rs_batch_prepare_launch.argtypes = [C.c_void_p, C.c_int32]
    L.rs_batch_upload_cqi_epochs_prb.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32]
    L.rs_batch_set_trace_prb.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.rs_batch_set_bearers.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
    L.rs_batch_set_arrivals.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]
    L.rs_batch_read_bearer_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rs_internet_flow_arrivals.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int32, C.POINTER(C.c_double),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.rs_get_rbg_size.argtypes = [C.c_int]
    L.rs_dl_prbs_for_bandwidth.argtypes = [C.c_double]
    L.rs_batch_download_cqi_epochs.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint8)]
    L.rs_batch_set_trace.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_int32)]
    L.rs_batch_run.argtypes = [C.c_void_p, C.c_int32]
    L.rs_batch_run_async.argtypes = [C.c_void_p, C.c_int32]
    L.rs_batch_sync.argtypes = [C.c_void_p]
    L.rs_batch_run_logged.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int16), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.POINTER(C.c_int32)]
    L.rs_batch_run_timed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    L.rs_batch_read_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.rs_batch_write_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rs_batch_slice_bytes_device.argtypes = [C.c_void_p, C.c_void_p]
    L.rs_batch_slice_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.rs_jit_selfcheck.argtypes = [C.c_int] * 6 + [C.c_char_p, C.c_size_t]
    L.rs_jit_selfcheck_queue.argtypes = [C.c_int] * 6 + [C.c_char_p, C.c_size_t]
    L.rs_batch_debug_stamps.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64)]
    L.rs_batch_debug_heap_sorts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.rs_ctx_debug_heap_sorts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.rs_jit_cache_stats.argtypes = [C.POINTER(C.c_longlong)]
    L.rs_jit_cache_stats.restype = None
    L.rs_jit_cache_file.argtypes = [C.c_int] * 7 + [C.c_char_p, C.c_size_t]
    L.rs_jit_cache_warm.argtypes = [C.c_int] * 7 + [C.c_char_p, C.c_size_t]
    L.rs_batch_autotune_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.rs_batch_debug_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rs_batch_checkpoint_bytes.restype = C.c_int64
    L.rs_batch_checkpoint_bytes.argtypes = [C.c_void_p]
    L.rs_batch_checkpoint_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rs_batch_checkpoint_load.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.rs_batch_ttis_done.restype = C.c_int64
    L.rs_batch_ttis_done.argtypes = [C.c_void_p]
    L.rs_batch_stream.restype = C.c_void_p
    L.rs_batch_stream.argtypes = [C.c_void_p]
    L.rs_batch_kernel_name.restype = C.c_char_p
    L.rs_batch_kernel_name.argtypes = [C.c_void_p]
    L.rs_lds_bytes_per_cell.argtypes = [C.c_int] * 5
    L.rs_device_source_hash.restype = C.c_char_p
    L.rs_device_source_hash.argtypes = []
    L.rs_hbm_copy_probe.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    L.rs_trace_read_mapping.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.c_int32]
    L.rs_trace_read_ue_log.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8),
                                       C.POINTER(C.c_uint8)]
    L.rs_trace_load_dir.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RadioSaberError(rc, lib().rs_last_error().decode())


def _count(rc):
    """calls that return a count (>= 0) or a negative RS_ERR_*"""
    if rc < 0:
        raise RadioSaberError(rc, lib().rs_last_error().decode())
    return rc


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def jit_selfcheck(n_slices, n_users, n_rbgs, rbg_size, threads=512, sched=RS_SCHED_MAXCELL, queues=False, untuned=False, dropin=False):
    """Compile the shape-specialised kernel for one shape (hiprtc, no GPU needed); returns the code size.  queues=True: the
    queue-model kernel of the shape.  untuned=True: without the -mllvm tuning options (the library's fallback build)."""
    buf = C.create_string_buffer(4096)
    fn = lib().rs_jit_selfcheck_queue if queues else (lib().rs_jit_selfcheck_untuned if untuned else lib().rs_jit_selfcheck)
    if dropin:  # the drop-in entry point's one-TTI kernel of a context of this shape (rs_ctx_specialize)
        fn = lib().rs_jit_selfcheck_dropin
    n = fn(n_slices, n_users, n_rbgs, rbg_size, threads, sched, buf, 4096)
    if n < 0:
        raise RadioSaberError(n, buf.value.decode(errors="replace"))
    return n


def jit_cache_stats():
    """This process's disk-cache counters of the run-time compiled kernels: dict(hits, misses, stores, rejected)."""
    out = (C.c_longlong * 4)()
    lib().rs_jit_cache_stats(out)
    return dict(zip(("hits", "misses", "stores", "rejected"), (int(x) for x in out)))


def jit_cache_file(n_slices, n_users, n_rbgs, rbg_size, threads=512, sched=RS_SCHED_MAXCELL, lean=False, streamed=False):
    """Path of the cache file the batch kernel of this shape lives in ('' when no cache directory can be named)."""
    buf = C.create_string_buffer(4096)
    lib().rs_jit_cache_file(n_slices, n_users, n_rbgs, rbg_size, threads, sched, (4 if lean else 0) | (2 if streamed else 0), buf, 4096)
    return buf.value.decode()


def jit_cache_warm(n_slices, n_users, n_rbgs, rbg_size, threads=512, sched=RS_SCHED_MAXCELL, lean=False, streamed=False):
    """Compile (or load) the batch kernel of this shape through the disk cache; no GPU needed.  Returns the code size."""
    buf = C.create_string_buffer(4096)
    n = lib().rs_jit_cache_warm(n_slices, n_users, n_rbgs, rbg_size, threads, sched, (4 if lean else 0) | (2 if streamed else 0), buf, 4096)
    if n < 0:
        raise RadioSaberError(n, buf.value.decode(errors="replace"))
    return n


def device_count():
    return lib().rs_device_count()


def device_source_hash():
    """Identity of the device code inside the library (FNV-1a of the embedded kernel sources, 16 hex digits)."""
    return lib().rs_device_source_hash().decode()


def get_rbg_size(nb_rbs):
    """PRBs per RBG (ref: src/utility/eesm-effective-sinr.h:82-103); raises above 512 PRBs like the reference throws."""
    return _count(lib().rs_get_rbg_size(nb_rbs))


def dl_prbs_for_bandwidth(bw_mhz):
    """PRBs of a downlink bandwidth in MHz (ref: src/core/spectrum/bandwidth-manager.cpp:30-38, 52-108)."""
    return lib().rs_dl_prbs_for_bandwidth(float(bw_mhz))


BEARER_NONE, BEARER_BACKLOG, BEARER_QUEUE = 0, 1, 2
FULL_PACKET = 1495  # MAXMTUSIZE 1490 + UDP 8 + IP 20, ROHC 28 -> 3, PDCP 2 (ref: src/protocolStack/packet/Packet.cpp:84-118)


def internet_flow_arrivals(rate_mbps, start_time, stop_time, size_seed, max_bursts=1 << 16):
    """Arrival bursts of one InternetFlow application (ref: src/flows/application/InternetFlow.cpp): (time f64[n], n_full i32[n],
    last_bytes i32[n]); no GPU needed."""
    t = np.zeros(max_bursts, np.float64)
    nf = np.zeros(max_bursts, np.int32)
    la = np.zeros(max_bursts, np.int32)
    n = _count(lib().rs_internet_flow_arrivals(rate_mbps, start_time, stop_time, size_seed, max_bursts, _p(t, C.c_double),
                                               _p(nf, C.c_int32), _p(la, C.c_int32)))
    return t[:n].copy(), nf[:n].copy(), la[:n].copy()


def frames_to_bursts(times, frame_bytes, mtu=1490):
    """Frames of a video trace as arrival bursts (ref: src/flows/application/TraceBased.cpp:155-224): size/1490 packets of
    1490 + 5 bytes and, for a remainder, one of remainder + 5 bytes (this application adds the headers to the last packet too)."""
    fb = np.asarray(frame_bytes, np.int64)
    rem = fb % mtu
    return (np.ascontiguousarray(times, np.float64), (fb // mtu).astype(np.int32),
            np.where(rem > 0, rem + (FULL_PACKET - mtu), 0).astype(np.int32))


def lds_bytes_per_cell(n_slices, n_users, n_rbgs, sched=RS_SCHED_MAXCELL, threads=512):
    """LDS bytes of one cell of this shape (<= 40 960: four cells per CU, <= 81 920: two)."""
    return _count(lib().rs_lds_bytes_per_cell(n_slices, n_users, n_rbgs, sched, threads))


def hbm_copy_probe(device=0, nbytes=1 << 30, iters=10):
    """GB/s of a 16 B/lane streaming copy (read + write bytes over time) -- the attainable HBM rate."""
    g = C.c_double(0)
    _check(lib().rs_hbm_copy_probe(device, nbytes, iters, C.byref(g)))
    return g.value


def link_tables(pinned=False):
    """Link adaptation tables (no GPU needed): dict of eff/kbps/E/X, each float64[16] -- as this host's libm evaluates them, or
    (pinned=True) the glibc-2.35 set compiled into the library (rs_config.link_tables)."""
    out = [np.zeros(16, np.float64) for _ in range(4)]
    _check((lib().rs_link_tables_pinned if pinned else lib().rs_link_tables)(*[_p(a, C.c_double) for a in out]))
    return dict(zip(("eff", "kbps", "eesm_e", "eesm_x"), out))


def link_tables_compare():
    """(n, text): how many EESM constants of this host's libm differ from the pinned glibc-2.35 set, and which."""
    buf = C.create_string_buffer(4096)
    return _count(lib().rs_link_tables_compare(buf, 4096)), buf.value.decode(errors="replace")


def jit_compiler_identity():
    """The compiler identity inside every cache key of the run-time builds (hiprtc / HIP runtime / clang with its LLVM commit / comgr)."""
    return lib().rs_jit_compiler_identity().decode()


def read_trace_mapping(path, max_entries=4096):
    """mapping<i>.config of the reference's cqi-traces-noise0 -> int32[n]: user u replays trace map[u % n]
    (ref: enb-mac-entity.cc:47-53, :171)."""
    out = np.zeros(max_entries, dtype=np.int32)
    n = _count(lib().rs_trace_read_mapping(os.fsencode(str(path)), _p(out, C.c_int32), max_entries))
    if n > max_entries:
        raise RadioSaberError(-1, f"{path}: {n} entries, max_entries={max_entries}")
    return out[:n].copy()


def read_ue_trace(path, nb_rbs=512, rbg_size=8, n_rows=475, per_prb=False):
    """One ue<id>.log -> (uint8[n_rows][nb_rbs/rbg_size], mixed) or, per_prb=True, (uint8[n_rows][nb_rbs], mixed);
    `mixed` counts RBGs whose PRBs differ (0: the RBG-granular replay is exact).  ref: enb-mac-entity.cc:173-186."""
    R = nb_rbs // rbg_size
    rbg = np.zeros((n_rows, R), dtype=np.uint8)
    prb = np.zeros((n_rows, nb_rbs), dtype=np.uint8) if per_prb else None
    rc = _count(lib().rs_trace_read_ue_log(os.fsencode(str(path)), n_rows, nb_rbs, rbg_size, _p(rbg, C.c_uint8),
                                           _p(prb, C.c_uint8) if per_prb else None))
    return (prb if per_prb else rbg), rc


def load_trace_dir(directory, n_traces=158, nb_rbs=512, rbg_size=8, n_rows=475):
    """ue0.log .. ue<n_traces-1>.log -> (uint8[n_traces][n_rows][R], mixed): the `trace` argument of
    BatchScheduler.set_trace (MAX_UE_TRACE = 158, MAX_TTI_TRACE = 475 in the reference)."""
    R = nb_rbs // rbg_size
    out = np.zeros((n_traces, n_rows, R), dtype=np.uint8)
    rc = _count(lib().rs_trace_load_dir(os.fsencode(str(directory)), n_traces, n_rows, nb_rbs, rbg_size,
                                        _p(out, C.c_uint8)))
    return out, rc


@dataclass
class SliceConfig:
    """The reference's JSON scheduler config (downlink-transport-scheduler.cpp:65-88;
    single-cell-with-interference.h:220-248): `slices` groups expand in order."""
    ues_per_slice: List[int]
    weight: List[float] = field(default_factory=list)
    algo_alpha: List[int] = field(default_factory=list)
    algo_beta: List[int] = field(default_factory=list)
    algo_epsilon: List[int] = field(default_factory=list)
    algo_psi: List[int] = field(default_factory=list)
    # per slice, what every UE of the slice runs (single-cell-with-interference.h:226-248, 300-440): {"backlog_flow": n,
    # "internet_flow": n, "if_bitrate": [Mbps per slice, ...], "video_app": n, "video_bitrate": [kbps, ...]}; empty = one
    # InfiniteBuffer flow per UE
    traffic: List[dict] = field(default_factory=list)

    def __post_init__(self):
        S = len(self.ues_per_slice)
        if not self.weight:
            self.weight = [1.0 / S] * S
        for name, default in (("algo_alpha", 0), ("algo_beta", 0), ("algo_epsilon", 1), ("algo_psi", 1)):
            if not getattr(self, name):
                setattr(self, name, [default] * S)
        for name in ("weight", "algo_alpha", "algo_beta", "algo_epsilon", "algo_psi"):
            if len(getattr(self, name)) != S:
                raise ValueError(f"{name} has {len(getattr(self, name))} entries for {S} slices")

    @classmethod
    def from_json(cls, path_or_dict):
        obj = path_or_dict if isinstance(path_or_dict, dict) else json.loads(Path(path_or_dict).read_text())
        ues = [int(x) for x in obj["ues_per_slice"]]
        w, a, b, e, p, tr = [], [], [], [], [], []
        for grp in obj["slices"]:
            for _ in range(int(grp["n_slices"])):
                w.append(float(grp["weight"]))
                a.append(int(grp.get("algo_alpha", 0)))
                b.append(int(grp.get("algo_beta", 0)))
                e.append(int(grp.get("algo_epsilon", 0)))
                p.append(int(grp.get("algo_psi", 0)))
                tr.append({k: grp[k] for k in ("backlog_flow", "internet_flow", "if_bitrate", "video_app", "video_bitrate")
                           if k in grp})
        return cls(ues, w, a, b, e, p, tr if any(tr) else [])

    def bearer_kinds(self):
        """[U][2] bearer kinds (index = priority) from `traffic`: InternetFlow j has priority j, every other application
        priority 0 (RadioBearer::GetPriority, src/flows/radio-bearer.cpp:88-97)."""
        k = np.zeros((self.n_users, 2), np.uint8)
        u2s = self.user_to_slice
        for u in range(self.n_users):
            t = self.traffic[u2s[u]] if self.traffic else {}
            nif = int(t.get("internet_flow", 0))
            if nif > 2:
                raise ValueError("more than MAX_BEARERS = 2 internet flows per UE")
            for j in range(nif):
                k[u, j] = BEARER_QUEUE
            if int(t.get("video_app", 0)) and not nif:
                k[u, 0] = BEARER_QUEUE
            if (int(t.get("backlog_flow", 0)) or not t) and not k[u, 0]:
                k[u, 0] = BEARER_BACKLOG
        return k

    @property
    def n_slices(self):
        return len(self.ues_per_slice)

    @property
    def n_users(self):
        return int(sum(self.ues_per_slice))

    @property
    def user_to_slice(self):
        return np.repeat(np.arange(self.n_slices, dtype=np.int32), self.ues_per_slice).astype(np.int32)


class _CfgHolder:
    """Keeps the numpy arrays a C rs_config points at alive."""

    def __init__(self, slices: SliceConfig, n_rbgs, rbg_size, sched, device, stream, synthetic_exp=False, link_tables=0):
        self.w = np.ascontiguousarray(slices.weight, np.float64)
        self.a = np.ascontiguousarray(slices.algo_alpha, np.int32)
        self.b = np.ascontiguousarray(slices.algo_beta, np.int32)
        self.e = np.ascontiguousarray(slices.algo_epsilon, np.int32)
        self.p = np.ascontiguousarray(slices.algo_psi, np.int32)
        self.u2s = np.ascontiguousarray(slices.user_to_slice, np.int32)
        self.c = _Config(slices.n_slices, slices.n_users, n_rbgs, rbg_size, sched, device,
                         _p(self.w, C.c_double), _p(self.a, C.c_int32), _p(self.b, C.c_int32),
                         _p(self.e, C.c_int32), _p(self.p, C.c_int32), _p(self.u2s, C.c_int32),
                         C.c_void_p(stream or 0), int(bool(synthetic_exp)), int(link_tables))


@dataclass
class TtiResult:
    target_rbs: np.ndarray
    quota_rbgs: np.ndarray
    rbg_to_user: np.ndarray
    user_nprb: np.ndarray
    user_final_cqi: np.ndarray
    user_mcs: np.ndarray
    user_tbs_bits: np.ndarray
    upper_rbg: Optional[np.ndarray] = None   # RS_SCHED_UPPERBOUND: [S][R] RBGs each slice took, push order, -1 padded
    upper_user: Optional[np.ndarray] = None  # ... and the user each one went to


class TtiScheduler:
    """Drop-in mode: RBsAllocation() of one TTI on the GPU (rs_create / rs_schedule_tti)."""

    def __init__(self, slices: SliceConfig, n_rbgs: int, rbg_size: int, sched: int = RS_SCHED_MAXCELL,
                 device: int = 0, stream: Optional[int] = None, synthetic_exp: bool = False, jit: bool = False,
                 link_tables: int = RS_LINK_DEFAULT):
        """jit: rs_ctx_specialize -- this context's own hiprtc build of the one-TTI kernel (identical results, shorter calls; its first
        calls run beside the built-in kernel unless the build carries the self-check mark: jit_status()).
        link_tables: RS_LINK_* (default for a drop-in context: this host's libm; create_warning says when it is not the fixtures')."""
        self.slices, self.R, self.rbg_size, self.sched = slices, n_rbgs, rbg_size, sched
        self._cfg = _CfgHolder(slices, n_rbgs, rbg_size, sched, device, stream, synthetic_exp, link_tables)
        self._h = lib().rs_create_checked(C.byref(self._cfg.c), RS_ABI_VERSION, C.sizeof(_Config))
        if not self._h:
            raise RadioSaberError(-1, lib().rs_last_error().decode())
        self.create_warning = lib().rs_last_error().decode()
        if jit:
            _check(lib().rs_ctx_specialize(self._h))

    def jit_status(self):
        """(code, message) of rs_ctx_jit_status: 1 specialised kernels serve, 0 not asked for, -1 build failed, -2 dropped by the self-check."""
        buf = C.create_string_buffer(512)
        rc = lib().rs_ctx_jit_status(self._h, buf, 512)
        return rc, buf.value.decode(errors="replace")

    def close(self):
        if getattr(self, "_h", None):
            lib().rs_destroy(self._h)
            self._h = None

    __del__ = close

    def schedule_tti(self, cqi, avg_rate, rand0=0, rand1=0, user_id: Optional[Sequence[int]] = None,
                     cqi_prb=None, hol_delay=None, prio_has_data=None, rand_draws=None, required_rbs=None,
                     data_to_transmit=None, cqi_epoch: int = 0) -> TtiResult:
        """cqi [n][R] per-RBG CQI, or cqi_prb [n][R*rbg_size] per-PRB CQI (then cqi may be None).
        rand_draws (RS_SCHED_NVS_NONGREEDY): the 300 * n rand() values of RBsAllocationNonGreedyPF, in draw order.
        cqi_epoch: non-zero = the caller's version number of the CQI block; a call with the number (and users) of the call before reads
        the context's device-resident image instead of the block (rs_tti_in.cqi_epoch)."""
        prb = None
        if cqi_prb is not None:
            prb = np.ascontiguousarray(cqi_prb, np.uint8)
            n = prb.shape[0]
            assert prb.shape == (n, self.R * self.rbg_size)
            cqi = None
        else:
            cqi = np.ascontiguousarray(cqi, np.uint8)
            n = cqi.shape[0]
            assert cqi.shape == (n, self.R)
        avg = np.ascontiguousarray(avg_rate, np.float64)
        assert avg.shape == (n,)
        uid = None if user_id is None else np.ascontiguousarray(user_id, np.int32)
        hol = None if hol_delay is None else np.ascontiguousarray(hol_delay, np.float64)
        prio = None if prio_has_data is None else np.ascontiguousarray(prio_has_data, np.uint8)
        draws = None if rand_draws is None else np.ascontiguousarray(rand_draws, np.int32)
        assert draws is None or draws.size == 300 * n
        req = None if required_rbs is None else np.ascontiguousarray(required_rbs, np.int32)
        dat = None if data_to_transmit is None else np.ascontiguousarray(data_to_transmit, np.int32)
        assert (req is None or req.shape == (n,)) and (dat is None or dat.shape == (n,))
        S = self.slices.n_slices
        res = TtiResult(np.zeros(S, np.int32), np.zeros(S, np.int32), np.zeros(self.R, np.int32),
                        np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32))
        tin = _TtiIn(n, _p(uid, C.c_int32) if uid is not None else None,
                     _p(cqi, C.c_uint8) if cqi is not None else None, _p(avg, C.c_double), rand0, rand1,
                     _p(prb, C.c_uint8) if prb is not None else None,
                     _p(hol, C.c_double) if hol is not None else None,
                     _p(prio, C.c_uint8) if prio is not None else None,
                     _p(draws, C.c_int32) if draws is not None else None,
                     _p(req, C.c_int32) if req is not None else None,
                     _p(dat, C.c_int32) if dat is not None else None, int(cqi_epoch))
        if self.sched == RS_SCHED_UPPERBOUND:
            res.upper_rbg = np.full((S, self.R), -1, np.int32)
            res.upper_user = np.full((S, self.R), -1, np.int32)
        tout = _TtiOut(_p(res.target_rbs, C.c_int32), _p(res.quota_rbgs, C.c_int32),
                       _p(res.rbg_to_user, C.c_int32), _p(res.user_nprb, C.c_int32),
                       _p(res.user_final_cqi, C.c_int32), _p(res.user_mcs, C.c_int32),
                       _p(res.user_tbs_bits, C.c_int32),
                       _p(res.upper_rbg, C.c_int32) if res.upper_rbg is not None else None,
                       _p(res.upper_user, C.c_int32) if res.upper_user is not None else None)
        _check(lib().rs_schedule_tti(self._h, C.byref(tin), C.byref(tout)))
        return res

    def heap_sorts(self):
        """int64[3]: heap-sort fallbacks of the std::sort emulation so far, per device site (rs_ctx_debug_heap_sorts)."""
        out = np.zeros(3, np.int64)
        _check(lib().rs_ctx_debug_heap_sorts(self._h, _p(out, C.c_int64)))
        return out

    @property
    def slice_offset(self):
        out = np.zeros(self.slices.n_slices, np.float64)
        _check(lib().rs_get_slice_offset(self._h, _p(out, C.c_double)))
        return out

    @slice_offset.setter
    def slice_offset(self, v):
        a = np.ascontiguousarray(v, np.float64)
        _check(lib().rs_set_slice_offset(self._h, _p(a, C.c_double)))


class BatchScheduler:
    """Many independent cells resident on one MI355X (rs_batch_*)."""

    def __init__(self, slices: SliceConfig, n_rbgs: int, rbg_size: int, n_cells: int,
                 sched: int = RS_SCHED_MAXCELL, device: int = 0, first_tti: int = 100, cqi_refresh: int = 40,
                 phy_error_draws: bool = False, threads_per_cell: int = 0, stream: Optional[int] = None,
                 jit: bool = False, synthetic_exp: bool = False, cqi_epoch_wrap: bool = False, queue_state_lds: int = 0,
                 autotune: bool = False, selfcheck=0, link_tables: int = RS_LINK_DEFAULT):
        """selfcheck: 0 / False = run-time builds without the self-check mark are checked against the built-in kernels before they serve
        (the default since ABI 11), 1 / True = every build, -1 = never.  link_tables: RS_LINK_* (default for batches: pinned glibc 2.35).
        synthetic_exp: the reference built with FIRST/SECOND_SYNTHETIC_EXP (transport blocks PRB by PRB; rs_config.synthetic_exp).
        cqi_epoch_wrap: the uploaded / synthesized epochs cycle instead of ending the run.  queue_state_lds: 0 auto, 1 LDS, -1 HBM."""
        self.slices, self.R, self.rbg_size, self.sched, self.n_cells = slices, n_rbgs, rbg_size, sched, n_cells
        self.S, self.U = slices.n_slices, slices.n_users
        self._cfg = _CfgHolder(slices, n_rbgs, rbg_size, sched, device, stream, synthetic_exp, link_tables)
        bc = _BatchConfig(self._cfg.c, n_cells, first_tti, cqi_refresh, int(phy_error_draws), threads_per_cell,
                          int(jit), int(bool(cqi_epoch_wrap)), int(queue_state_lds), int(bool(autotune)), int(selfcheck))
        self._h = lib().rs_batch_create_checked(C.byref(bc), RS_ABI_VERSION, C.sizeof(_BatchConfig))
        if not self._h:
            raise RadioSaberError(-1, lib().rs_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().rs_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def seed(self, seeds, rand_skip=None):
        s = np.ascontiguousarray(seeds, np.uint32)
        assert s.shape == (self.n_cells,)
        k = None if rand_skip is None else np.ascontiguousarray(rand_skip, np.int64)
        _check(lib().rs_batch_seed(self._h, _p(s, C.c_uint32), _p(k, C.c_int64) if k is not None else None))

    def upload_cqi_epochs(self, cqi):
        a = np.ascontiguousarray(cqi, np.uint8)
        assert a.ndim == 4 and a.shape[0] == self.n_cells and a.shape[2:] == (self.U, self.R), a.shape
        _check(lib().rs_batch_upload_cqi_epochs(self._h, _p(a, C.c_uint8), a.shape[1]))
        self.n_epochs = a.shape[1]

    def upload_cqi_epochs_prb(self, cqi_prb):
        """[n_cells][n_epochs][U][R*rbg_size]: per-PRB reports (the metric reads each RBG's first PRB, link adaptation all)."""
        a = np.ascontiguousarray(cqi_prb, np.uint8)
        assert a.ndim == 4 and a.shape[0] == self.n_cells and a.shape[2:] == (self.U, self.R * self.rbg_size), a.shape
        _check(lib().rs_batch_upload_cqi_epochs_prb(self._h, _p(a, C.c_uint8), a.shape[1]))
        self.n_epochs = a.shape[1]

    def set_trace_prb(self, trace_prb, user_trace, row_modulus=475):
        t = np.ascontiguousarray(trace_prb, np.uint8)
        assert t.ndim == 3 and t.shape[2] == self.R * self.rbg_size
        ut = np.ascontiguousarray(user_trace, np.int32)
        assert ut.shape == (self.n_cells, self.U)
        _check(lib().rs_batch_set_trace_prb(self._h, _p(t, C.c_uint8), t.shape[0], t.shape[1], row_modulus, _p(ut, C.c_int32)))

    def synthesize_cqi(self, seed, n_epochs, weights=TRACE_CQI_HISTOGRAM, first_cell=0):
        """Grids drawn on the device, keyed by (seed, first_cell + local cell, epoch, user, rbg)."""
        w = np.ascontiguousarray(weights, np.float64)
        assert w.shape == (15,)
        _check(lib().rs_batch_synthesize_cqi_at(self._h, seed, _p(w, C.c_double), n_epochs, first_cell))
        self.n_epochs = n_epochs

    def download_cqi_epochs(self, cell):
        out = np.zeros((self.n_epochs, self.U, self.R), np.uint8)
        _check(lib().rs_batch_download_cqi_epochs(self._h, cell, _p(out, C.c_uint8)))
        return out

    def set_trace(self, trace, user_trace, row_modulus=475):
        t = np.ascontiguousarray(trace, np.uint8)
        assert t.ndim == 3 and t.shape[2] == self.R
        ut = np.ascontiguousarray(user_trace, np.int32)
        assert ut.shape == (self.n_cells, self.U)
        _check(lib().rs_batch_set_trace(self._h, _p(t, C.c_uint8), t.shape[0], t.shape[1], row_modulus,
                                        _p(ut, C.c_int32)))

    def run(self, n_ttis):
        _check(lib().rs_batch_run(self._h, n_ttis))

    def run_async(self, n_ttis):
        _check(lib().rs_batch_run_async(self._h, n_ttis))

    def sync(self):
        _check(lib().rs_batch_sync(self._h))

    def run_logged(self, n_ttis, slice_keys=False):
        """slice_keys=True (transport schedulers) adds what the inter-slice step read: 'slice_cqi' [cells][ttis][R][S]
        (CQI of the slice's best user, 0 = no user) and 'slice_user' (its id, -1 = none)."""
        m = np.zeros((self.n_cells, n_ttis, self.R), np.int16)
        tb = np.zeros((self.n_cells, n_ttis, self.U), np.int32)
        q = np.zeros((self.n_cells, n_ttis, self.S), np.int16)
        tg = np.zeros((self.n_cells, n_ttis, self.S), np.int16)
        ui = np.zeros((self.n_cells, n_ttis, self.U), np.int32)
        keys = np.zeros((self.n_cells, n_ttis, self.R, self.S), np.uint32) if slice_keys else None
        lg = _BatchLog(_p(m, C.c_int16), _p(tb, C.c_int32), _p(q, C.c_int16), _p(tg, C.c_int16), _p(ui, C.c_int32),
                       _p(keys, C.c_uint32) if slice_keys else None)
        _check(lib().rs_batch_run_logged_ex(self._h, n_ttis, C.byref(lg)))
        out = {"rbg_to_user": m, "tbs_bits": tb, "quota": q, "target": tg, "nprb": ui & 0xFFFF,
               "final_cqi": (ui >> 16) & 0xFF, "mcs": (ui >> 24) & 0xFF}
        if slice_keys:
            out["slice_cqi"] = (keys & 0xFF).astype(np.int32)
            out["slice_user"] = (keys >> 8).astype(np.int32) - 1
        return out

    def run_timed(self, n_ttis, launches):
        ms = np.zeros(launches, np.float32)
        _check(lib().rs_batch_run_timed(self._h, n_ttis, launches, _p(ms, C.c_float)))
        return ms

    def state(self):
        avg = np.zeros((self.n_cells, self.U), np.float64)
        cb = np.zeros((self.n_cells, self.U), np.int64)
        cr = np.zeros((self.n_cells, self.U), np.int64)
        sl = np.zeros((self.n_cells, self.S), np.float64)
        _check(lib().rs_batch_read_state(self._h, _p(avg, C.c_double), _p(cb, C.c_int64), _p(cr, C.c_int64),
                                         _p(sl, C.c_double)))
        return {"avg_rate": avg, "cum_bytes": cb, "cum_rbs": cr, "slice_state": sl}

    def write_state(self, avg_rate=None, slice_state=None):
        """Set the PF averages [n_cells][U] (>= 1) and / or the slice state [n_cells][S] between launches (rs_batch_write_state)."""
        a = None if avg_rate is None else np.ascontiguousarray(avg_rate, np.float64)
        s = None if slice_state is None else np.ascontiguousarray(slice_state, np.float64)
        assert a is None or a.shape == (self.n_cells, self.U)
        assert s is None or s.shape == (self.n_cells, self.S)
        _check(lib().rs_batch_write_state(self._h, _p(a, C.c_double) if a is not None else None,
                                          _p(s, C.c_double) if s is not None else None))

    # ---- finite queues ----
    def set_bearers(self, bearer_kind):
        """bearer_kind [U][2] (index = bearer priority): BEARER_NONE / BEARER_BACKLOG / BEARER_QUEUE."""
        k = np.ascontiguousarray(bearer_kind, np.uint8)
        assert k.shape == (self.U, 2)
        _check(lib().rs_batch_set_bearers(self._h, _p(k, C.c_uint8)))
        self.bearer_kind = k

    def set_arrivals(self, bursts):
        """bursts[(cell, user, prio)] = (time f64[n], n_full i32[n], last_bytes i32[n]) for every finite-queue bearer that
        receives traffic (missing keys: no arrivals)."""
        nb = self.n_cells * self.U * 2
        off = np.zeros(nb + 1, np.int64)
        for (c, u, k), (t, _, _) in bursts.items():
            off[(c * self.U + u) * 2 + k + 1] = len(t)
        off = np.cumsum(off)
        total = int(off[-1])
        t_all = np.zeros(max(total, 1), np.float64)
        nf_all = np.zeros(max(total, 1), np.int32)
        la_all = np.zeros(max(total, 1), np.int32)
        for (c, u, k), (t, nf, la) in bursts.items():
            i = int(off[(c * self.U + u) * 2 + k])
            t_all[i:i + len(t)], nf_all[i:i + len(t)], la_all[i:i + len(t)] = t, nf, la
        _check(lib().rs_batch_set_arrivals(self._h, _p(off, C.c_int64), _p(t_all, C.c_double), _p(nf_all, C.c_int32),
                                           _p(la_all, C.c_int32)))

    def bearer_state(self):
        shp = (self.n_cells, self.U, 2)
        avg, cb, cr = np.zeros(shp, np.float64), np.zeros(shp, np.int64), np.zeros(shp, np.int64)
        qb, qp = np.zeros(shp, np.int32), np.zeros(shp, np.int32)
        _check(lib().rs_batch_read_bearer_state(self._h, _p(avg, C.c_double), _p(cb, C.c_int64), _p(cr, C.c_int64),
                                                _p(qb, C.c_int32), _p(qp, C.c_int32)))
        return {"avg_rate": avg, "cum_bytes": cb, "cum_rbs": cr, "queue_bytes": qb, "queue_packets": qp}

    def clock(self):
        """(t, last_update): the simulated time of the next TTI and RadioBearer::m_lastUpdate, per cell."""
        t = np.zeros(self.n_cells, np.float64)
        lu = np.zeros(self.n_cells, np.float64)
        _check(lib().rs_batch_read_clock(self._h, _p(t, C.c_double), _p(lu, C.c_double)))
        return t, lu

    def prepare_launch(self, n_ttis):
        """Build now the kernel an unlogged run(n_ttis) would build at its first launch (the lean build): keeps hiprtc out of timed runs."""
        _check(lib().rs_batch_prepare_launch(self._h, int(n_ttis)))

    def checkpoint(self):
        """Everything the batch carries from one launch to the next, as bytes (rs_batch_checkpoint_save)."""
        n = _count(lib().rs_batch_checkpoint_bytes(self._h))
        buf = C.create_string_buffer(n)
        _check(lib().rs_batch_checkpoint_save(self._h, buf, n))
        return buf.raw

    def restore(self, blob):
        """Continue from a checkpoint of a batch of the same shape (rs_batch_checkpoint_load); set the CQI source first."""
        _check(lib().rs_batch_checkpoint_load(self._h, blob, len(blob)))

    def debug_clocks(self):
        """(shader MHz [n_cells], ms [n_cells]) of the last launch, from the kernel's own two clocks (rs_batch_debug_clocks)."""
        mhz = np.zeros(self.n_cells, np.float64)
        ms = np.zeros(self.n_cells, np.float64)
        _check(lib().rs_batch_debug_clocks(self._h, _p(mhz, C.c_double), _p(ms, C.c_double)))
        return mhz, ms

    def autotune_report(self):
        """(candidates timed, text): what rs_batch_config.autotune measured and kept."""
        buf = C.create_string_buffer(1024)
        n = lib().rs_batch_autotune_report(self._h, buf, 1024)
        return n, buf.value.decode(errors="replace")

    def jit_status(self):
        """(code, message): 1 = shape-specialised kernel in use, 0 = not requested, -1 = requested but the build failed."""
        if not getattr(self, "_h", None):
            raise RadioSaberError(-4, "jit_status() of a closed batch")
        buf = C.create_string_buffer(512)
        rc = lib().rs_batch_jit_status(self._h, buf, 512)
        return rc, buf.value.decode(errors="replace")

    def slice_bytes(self):
        out = np.zeros(self.S, np.uint64)
        _check(lib().rs_batch_slice_bytes(self._h, _p(out, C.c_uint64)))
        return out

    def slice_bytes_into(self, device_ptr):
        """Reduce per-slice cumulative bytes into a device buffer (uint64[S]) on the batch's stream."""
        _check(lib().rs_batch_slice_bytes_device(self._h, C.c_void_p(device_ptr)))

    def heap_sorts(self):
        """int64[n_cells][3]: heap-sort fallbacks of the std::sort emulation so far, per cell and device site
        (0 workgroup level of the register form, 1 inside one wave's finish, 2 workgroup level of the LDS form)."""
        out = np.zeros((self.n_cells, 3), np.int64)
        _check(lib().rs_batch_debug_heap_sorts(self._h, _p(out, C.c_int64)))
        return out

    def debug_stamps(self, cell=0):
        out = np.zeros(20, np.uint64)
        _check(lib().rs_batch_debug_stamps(self._h, cell, _p(out, C.c_uint64)))
        return out

    @property
    def ttis_done(self):
        return lib().rs_batch_ttis_done(self._h)

    @property
    def stream(self):
        return lib().rs_batch_stream(self._h)

    @property
    def kernel_name(self):
        return lib().rs_batch_kernel_name(self._h).decode()
