"""radiosaber_amd -- MI355X-native RadioSaber downlink RBG allocation path.

Thin ctypes mirror of the C ABI in include/radiosaber_hip.h (libradiosaber_hip.so, hand-written
HIP for gfx950).  There is no CPU fallback: creating a scheduler without a HIP device, or without the
built library, raises.
"""
from . import toolchain as _toolchain

# the run-time kernels are built by the toolchain the library was built with, whatever is imported after this package (toolchain.py;
# a process that imported torch BEFORE this package keeps the wheel's compiler: call toolchain.prefer_system_compiler() first, as bench.py does)
_toolchain.prefer_system_compiler()

from .api import (  # noqa: F401,E402
    RS_SCHED_MAXCELL, RS_SCHED_NVS, RS_SCHED_PF, RS_SCHED_NVS_NONGREEDY, RS_SCHED_SEQUENTIAL, RS_SCHED_UPPERBOUND, RS_SCHED_VOGEL, RS_SCHED_SUBOPT,
    BEARER_BACKLOG, BEARER_NONE, BEARER_QUEUE, FULL_PACKET, TRACE_CQI_HISTOGRAM, frames_to_bursts, internet_flow_arrivals,
    BatchScheduler, RadioSaberError, SliceConfig, TtiResult, TtiScheduler, device_count, device_source_hash, dl_prbs_for_bandwidth, get_rbg_size, hbm_copy_probe, jit_selfcheck, lds_bytes_per_cell,
    jit_cache_file, jit_cache_stats, jit_cache_warm, jit_compiler_identity, lib, link_tables_compare,
    RS_LINK_DEFAULT, RS_LINK_HOST_LIBM, RS_LINK_PINNED_GLIBC_2_35,
    link_tables, load_trace_dir, read_trace_mapping, read_ue_trace,
)

__all__ = ["RS_SCHED_PF", "RS_SCHED_NVS", "RS_SCHED_SEQUENTIAL", "RS_SCHED_MAXCELL", "RS_SCHED_VOGEL", "RS_SCHED_SUBOPT", "RS_SCHED_UPPERBOUND", "RS_SCHED_NVS_NONGREEDY", "SliceConfig",
           "TtiScheduler", "TtiResult", "BatchScheduler", "RadioSaberError", "device_count", "lib",
           "link_tables", "jit_selfcheck", "device_source_hash", "TRACE_CQI_HISTOGRAM", "read_trace_mapping", "read_ue_trace",
           "load_trace_dir", "hbm_copy_probe", "lds_bytes_per_cell", "get_rbg_size", "dl_prbs_for_bandwidth", "internet_flow_arrivals", "frames_to_bursts",
           "BEARER_NONE", "BEARER_BACKLOG", "BEARER_QUEUE", "FULL_PACKET", "jit_cache_file", "jit_cache_stats", "jit_cache_warm",
           "jit_compiler_identity", "link_tables_compare", "RS_LINK_DEFAULT", "RS_LINK_HOST_LIBM", "RS_LINK_PINNED_GLIBC_2_35"]
