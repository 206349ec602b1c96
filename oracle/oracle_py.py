"""ctypes binding of the CPU oracle (oracle/librs_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (radiosaber_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = Path(os.environ.get("RS_ORACLE_LIB", HERE / "librs_oracle.so"))  # (tools/sanitize_cpu.sh points this at the ASan/UBSan build)
REF_DIR = HERE / "_ref"

SCHED_PF, SCHED_NVS, SCHED_SEQUENTIAL, SCHED_MAXCELL, SCHED_VOGEL, SCHED_UPPERBOUND, SCHED_NVS_NONGREEDY = 1, 7, 8, 9, 103, 10, 11
SCHED_SUBOPT = 101


def build(quiet=True):
    """(Re)build the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-C", str(HERE)], check=True,
                   stdout=subprocess.DEVNULL if quiet else None,
                   stderr=subprocess.DEVNULL if quiet else None)


class _Config(C.Structure):
    _fields_ = [("n_slices", C.c_int), ("n_users", C.c_int), ("n_rbgs", C.c_int),
                ("rbg_size", C.c_int), ("sched", C.c_int),
                ("weights", C.POINTER(C.c_double)), ("alpha", C.POINTER(C.c_int)),
                ("beta", C.POINTER(C.c_int)), ("epsilon", C.POINTER(C.c_int)),
                ("psi", C.POINTER(C.c_int)), ("user_to_slice", C.POINTER(C.c_int))]


class _TtiOut(C.Structure):
    _fields_ = [("target_rbs", C.POINTER(C.c_int)), ("quota_rbgs", C.POINTER(C.c_int)),
                ("rbg_to_user", C.POINTER(C.c_int)), ("user_nprb", C.POINTER(C.c_int)),
                ("user_final_cqi", C.POINTER(C.c_int)), ("user_mcs", C.POINTER(C.c_int)),
                ("user_tbs_bits", C.POINTER(C.c_int)), ("served_slice", C.c_int),
                ("upper_rbg", C.POINTER(C.c_int)), ("upper_user", C.POINTER(C.c_int)),
                ("slice_eff", C.POINTER(C.c_double)), ("slice_user", C.POINTER(C.c_int))]


class _TraceRun(C.Structure):
    _fields_ = [("trace", C.POINTER(C.c_uint8)), ("n_traces", C.c_int), ("n_rows", C.c_int),
                ("mapping", C.POINTER(C.c_int)), ("n_map", C.c_int), ("row_modulus", C.c_int),
                ("seed", C.c_uint), ("rand_skip", C.c_long), ("phy_error_draws", C.c_int),
                ("first_tti", C.c_int), ("n_ttis", C.c_int)]


class _Rng(C.Structure):
    _fields_ = [("r", C.c_int32 * 34), ("f", C.c_int), ("b", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        L = C.CDLL(str(LIB))
        L.rso_tbs_table.restype = C.POINTER(C.c_int)
        L.rso_mcs_to_itbs.restype = C.POINTER(C.c_int)
        L.rso_cqi_to_mcs.restype = C.POINTER(C.c_int)
        L.rso_sinr_for_cqi.restype = C.POINTER(C.c_double)
        L.rso_efficiency_from_cqi.restype = C.c_double
        L.rso_efficiency_from_cqi.argtypes = [C.c_int]
        L.rso_cqi_from_sinr.argtypes = [C.c_double]
        L.rso_tbs_bits.argtypes = [C.c_int, C.c_int]
        L.rso_eesm_effective_sinr.restype = C.c_double
        L.rso_eesm_effective_sinr.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.rso_final_cqi.argtypes = [C.POINTER(C.c_uint8), C.c_int]
        L.rso_cell_create.restype = C.c_void_p
        L.rso_cell_create.argtypes = [C.POINTER(_Config)]
        L.rso_cell_destroy.argtypes = [C.c_void_p]
        L.rso_cell_set_cqi.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.rso_cell_set_cqi_prb.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.rso_cell_set_queue_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8)]
        L.rso_cell_set_last_update.argtypes = [C.c_void_p, C.c_double]
        L.rso_cell_set_synthetic_exp.argtypes = [C.c_void_p, C.c_int]
        L.rso_cell_set_avg_rate.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.rso_cell_set_second_bearer_avg.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.rso_cell_step.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(_TtiOut)]
        L.rso_cell_allocate.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(_TtiOut)]
        L.rso_cell_allocate_nongreedy.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.c_int,
                                                  C.POINTER(_TtiOut)]
        L.rso_cell_get_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.rso_run_trace.argtypes = [C.c_void_p, C.POINTER(_TraceRun)] + [C.POINTER(C.c_int)] * 5
        L.rso_run_synth.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_uint,
                                    C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rso_clock_ticks.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.rso_run_synth_many.argtypes = [C.POINTER(_Config), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                         C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                         C.POINTER(C.c_int)]
        L.rso_run_synth_cells.argtypes = [C.POINTER(_Config), C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint), C.c_int,
                                          C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.rso_cell_enable_queues.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.rso_cell_set_arrivals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32)]
        L.rso_cell_step_queues.argtypes = [C.c_void_p, C.c_double, C.POINTER(_Rng), C.POINTER(_TtiOut)]
        L.rso_run_synth_queues.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_uint, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rso_run_synth_queues_prb.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_uint, C.c_int,
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rso_cell_get_bearer_state.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.rso_run_synth_prb.argtypes = L.rso_run_synth.argtypes
        L.rso_run_trace_prb.argtypes = L.rso_run_trace.argtypes
        L.rso_srand.argtypes = [C.POINTER(_Rng), C.c_uint]
        L.rso_rand.argtypes = [C.POINTER(_Rng)]
        for fn in ("rso_greedy_by_row", "rso_maximize_cell", "rso_vogel", "rso_subopt"):
            getattr(L, fn).argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_int,
                                       C.POINTER(C.c_int)]
        L.rso_maximize_cell_order.argtypes = [C.POINTER(C.c_double), C.c_int, C.c_int, C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _optp(a, t):
    return _p(a, t) if a is not None else None


def tables():
    L = lib()
    return {
        "tbs": np.ctypeslib.as_array(L.rso_tbs_table(), (110, 27)).copy(),
        "mcs_to_itbs": np.ctypeslib.as_array(L.rso_mcs_to_itbs(), (29,)).copy(),
        "cqi_to_mcs": np.ctypeslib.as_array(L.rso_cqi_to_mcs(), (15,)).copy(),
        "sinr_for_cqi": np.ctypeslib.as_array(L.rso_sinr_for_cqi(), (15,)).copy(),
    }


def eesm(sinr_db):
    a = np.ascontiguousarray(sinr_db, np.float64)
    return lib().rso_eesm_effective_sinr(_p(a, C.c_double), len(a))


def final_cqi(cqi_per_prb):
    a = np.ascontiguousarray(cqi_per_prb, np.uint8)
    return lib().rso_final_cqi(_p(a, C.c_uint8), len(a))


def clock_ticks(first_tti, n):
    """Simulated time at the start of scheduled TTIs first_tti .. first_tti + n - 1, as the oracle's run loops form it."""
    out = np.zeros(n, np.float64)
    lib().rso_clock_ticks(first_tti, n, _p(out, C.c_double))
    return out


class Rng:
    """glibc TYPE_3 rand() restatement."""

    def __init__(self, seed):
        self.g = _Rng()
        lib().rso_srand(C.byref(self.g), seed)

    def rand(self):
        return lib().rso_rand(C.byref(self.g))


def interslice(fn, eff, quota):
    """fn in {'greedy_by_row','maximize_cell','vogel','subopt'}; eff [R][S] float64 -> rbg_to_slice [R]."""
    eff = np.ascontiguousarray(eff, np.float64)
    R, S = eff.shape
    q = np.ascontiguousarray(quota, np.int32)
    out = np.empty(R, np.int32)
    getattr(lib(), "rso_" + fn)(_p(eff, C.c_double), _p(q, C.c_int), R, S, _p(out, C.c_int))
    return out


def maximize_cell_order(eff):
    eff = np.ascontiguousarray(eff, np.float64)
    R, S = eff.shape
    out = np.empty(R * S, np.int32)
    lib().rso_maximize_cell_order(_p(eff, C.c_double), R, S, _p(out, C.c_int))
    return out


class TtiOut:
    def __init__(self, S, U, R):
        self.target_rbs = np.zeros(S, np.int32)
        self.quota_rbgs = np.zeros(S, np.int32)
        self.rbg_to_user = np.zeros(R, np.int32)
        self.user_nprb = np.zeros(U, np.int32)
        self.user_final_cqi = np.zeros(U, np.int32)
        self.user_mcs = np.zeros(U, np.int32)
        self.user_tbs_bits = np.zeros(U, np.int32)
        self.upper_rbg = np.full((S, R), -1, np.int32)   # sched 10 only
        self.upper_user = np.full((S, R), -1, np.int32)
        self.slice_eff = np.zeros((R, S), np.float64)    # transport schedulers: what the inter-slice step reads
        self.slice_user = np.full((R, S), -1, np.int32)
        self.c = _TtiOut(_p(self.target_rbs, C.c_int), _p(self.quota_rbgs, C.c_int),
                         _p(self.rbg_to_user, C.c_int), _p(self.user_nprb, C.c_int),
                         _p(self.user_final_cqi, C.c_int), _p(self.user_mcs, C.c_int),
                         _p(self.user_tbs_bits, C.c_int), -1, _p(self.upper_rbg, C.c_int), _p(self.upper_user, C.c_int),
                         _p(self.slice_eff, C.c_double), _p(self.slice_user, C.c_int))

    @property
    def served_slice(self):
        return self.c.served_slice


class Cell:
    """One cell of the oracle.  ues_per_slice: list[int]; weights/eps/psi per slice."""

    def __init__(self, ues_per_slice, n_rbgs, rbg_size, sched, weights=None, epsilon=None, psi=None,
                 alpha=None, beta=None):
        S = len(ues_per_slice)
        self.S, self.R, self.rbg_size, self.sched = S, n_rbgs, rbg_size, sched
        self.u2s = np.repeat(np.arange(S, dtype=np.int32), ues_per_slice).astype(np.int32)
        self.U = len(self.u2s)
        self.weights = np.ascontiguousarray(weights if weights is not None else np.full(S, 1.0 / S), np.float64)
        self.eps = np.ascontiguousarray(epsilon if epsilon is not None else np.ones(S), np.int32)
        self.psi = np.ascontiguousarray(psi if psi is not None else np.ones(S), np.int32)
        self.alpha = np.ascontiguousarray(alpha if alpha is not None else np.zeros(S), np.int32)
        self.beta = np.ascontiguousarray(beta if beta is not None else np.zeros(S), np.int32)
        self.cfg = _Config(S, self.U, n_rbgs, rbg_size, sched, _p(self.weights, C.c_double),
                           _p(self.alpha, C.c_int), _p(self.beta, C.c_int), _p(self.eps, C.c_int),
                           _p(self.psi, C.c_int), _p(self.u2s, C.c_int))
        self.h = lib().rso_cell_create(C.byref(self.cfg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().rso_cell_destroy(self.h)
            self.h = None

    def new_out(self):
        return TtiOut(self.S, self.U, self.R)

    def set_cqi(self, cqi):
        a = np.ascontiguousarray(cqi, np.uint8)
        assert a.shape == (self.U, self.R)
        lib().rso_cell_set_cqi(self.h, _p(a, C.c_uint8))

    def set_cqi_prb(self, prb):
        a = np.ascontiguousarray(prb, np.uint8)
        assert a.shape == (self.U, self.R * self.rbg_size)
        lib().rso_cell_set_cqi_prb(self.h, _p(a, C.c_uint8))

    def set_synthetic_exp(self, on=True):
        lib().rso_cell_set_synthetic_exp(self.h, 1 if on else 0)

    def set_queue_state(self, hol, prio_has_data):
        h = np.ascontiguousarray(hol, np.float64)
        q = np.ascontiguousarray(prio_has_data, np.uint8)
        assert h.shape == (self.U,) and q.shape == (self.U,)
        lib().rso_cell_set_queue_state(self.h, _p(h, C.c_double), _p(q, C.c_uint8))

    def set_second_bearer_avg(self, avg2):
        """avg2[u] >= 0: user u holds a second bearer with that average rate this TTI; None clears."""
        if avg2 is None:
            lib().rso_cell_set_second_bearer_avg(self.h, None)
            return
        a = np.ascontiguousarray(avg2, np.float64)
        assert a.shape == (self.U,)
        lib().rso_cell_set_second_bearer_avg(self.h, _p(a, C.c_double))

    def set_slice_offset(self, offset):
        """slice_rbs_offset_ [S] from outside (a test that follows a context whose calls the oracle did not all take part in)."""
        a = np.ascontiguousarray(offset, np.float64)
        assert a.shape == (self.S,)
        lib().rso_cell_set_slice_offset.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib().rso_cell_set_slice_offset(self.h, _p(a, C.c_double))

    def set_avg_rate(self, avg):
        a = np.ascontiguousarray(avg, np.float64)
        lib().rso_cell_set_avg_rate(self.h, _p(a, C.c_double))

    def set_last_update(self, t):
        lib().rso_cell_set_last_update(self.h, t)

    def step(self, now, rand0, rand1, out):
        return lib().rso_cell_step(self.h, now, rand0, rand1, C.byref(out.c))

    def allocate(self, avg_rate, rand0, rand1, out):
        a = np.ascontiguousarray(avg_rate, np.float64)
        return lib().rso_cell_allocate(self.h, _p(a, C.c_double), rand0, rand1, C.byref(out.c))

    def allocate_nongreedy(self, avg_rate, slice_id, draws, out):
        """sched 11: RBsAllocationNonGreedyPF for the users of `slice_id` with the rand() values in draw order."""
        a = np.ascontiguousarray(avg_rate, np.float64)
        d = np.ascontiguousarray(draws, np.int32)
        return lib().rso_cell_allocate_nongreedy(self.h, _p(a, C.c_double), slice_id, _p(d, C.c_int), d.size, C.byref(out.c))

    # ---- finite queues (parity unpinned) ----
    def enable_queues(self, bearer_kind):
        """bearer_kind [U][2] (index = priority): 0 none, 1 InfiniteBuffer, 2 finite queue."""
        k = np.ascontiguousarray(bearer_kind, np.uint8)
        assert k.shape == (self.U, 2)
        lib().rso_cell_enable_queues(self.h, _p(k, C.c_uint8))

    def set_arrivals(self, user, prio, time, n_full, last):
        t = np.ascontiguousarray(time, np.float64)
        nf = np.ascontiguousarray(n_full, np.int32)
        la = np.ascontiguousarray(last, np.int32)
        assert t.shape == nf.shape == la.shape
        lib().rso_cell_set_arrivals(self.h, user, prio, len(t), _p(t, C.c_double), _p(nf, C.c_int32), _p(la, C.c_int32))

    def step_queues(self, now, rng, out):
        return lib().rso_cell_step_queues(self.h, now, C.byref(rng.g), C.byref(out.c))

    def run_synth_queues(self, cqi_epochs, seed, n_ttis, refresh=40, per_prb=False):
        e = np.ascontiguousarray(cqi_epochs, np.uint8)
        assert e.shape[1:] == (self.U, self.R * (self.rbg_size if per_prb else 1))
        logs = {"rbg_to_user": np.zeros((n_ttis, self.R), np.int32), "tbs_bits": np.zeros((n_ttis, self.U), np.int32)}
        fn = lib().rso_run_synth_queues_prb if per_prb else lib().rso_run_synth_queues
        rc = fn(self.h, _p(e, C.c_uint8), e.shape[0], refresh, seed, n_ttis,
                _p(logs["rbg_to_user"], C.c_int), _p(logs["tbs_bits"], C.c_int))
        if rc:
            raise RuntimeError(f"rso_run_synth_queues rc={rc}")
        return logs

    def bearer_state(self):
        avg = np.zeros((self.U, 2), np.float64)
        cb = np.zeros((self.U, 2), np.int64)
        cr = np.zeros((self.U, 2), np.int64)
        qb = np.zeros((self.U, 2), np.int32)
        qp = np.zeros((self.U, 2), np.int32)
        lib().rso_cell_get_bearer_state(self.h, _p(avg, C.c_double), _p(cb, C.c_int64), _p(cr, C.c_int64), _p(qb, C.c_int32),
                                        _p(qp, C.c_int32))
        return {"avg_rate": avg, "cum_bytes": cb, "cum_rbs": cr, "queue_bytes": qb, "queue_packets": qp}

    def state(self):
        avg = np.zeros(self.U, np.float64)
        cb = np.zeros(self.U, np.int64)
        cr = np.zeros(self.U, np.int64)
        sl = np.zeros(self.S, np.float64)
        lib().rso_cell_get_state(self.h, _p(avg, C.c_double), _p(cb, C.c_int64), _p(cr, C.c_int64),
                                 _p(sl, C.c_double))
        return {"avg_rate": avg, "cum_bytes": cb, "cum_rbs": cr, "slice_state": sl}

    def run_trace(self, trace, mapping, seed, rand_skip, n_ttis, phy_error_draws=1, first_tti=100,
                  row_modulus=475, log=True, per_prb=False):
        trace = np.ascontiguousarray(trace, np.uint8)
        mapping = np.ascontiguousarray(mapping, np.int32)
        n_tr, n_rows, R = trace.shape
        assert R == self.R * (self.rbg_size if per_prb else 1)
        run = _TraceRun(_p(trace, C.c_uint8), n_tr, n_rows, _p(mapping, C.c_int), len(mapping),
                        row_modulus, seed, rand_skip, phy_error_draws, first_tti, n_ttis)
        logs = None
        if log:
            logs = {"rbg_to_user": np.zeros((n_ttis, self.R), np.int32),
                    "final_cqi": np.zeros((n_ttis, self.U), np.int32),
                    "quota": np.zeros((n_ttis, self.S), np.int32),
                    "target": np.zeros((n_ttis, self.S), np.int32),
                    "tbs_bits": np.zeros((n_ttis, self.U), np.int32)}
        rc = (lib().rso_run_trace_prb if per_prb else lib().rso_run_trace)(self.h, C.byref(run),
                                 _optp(logs and logs["rbg_to_user"], C.c_int),
                                 _optp(logs and logs["final_cqi"], C.c_int),
                                 _optp(logs and logs["quota"], C.c_int),
                                 _optp(logs and logs["target"], C.c_int),
                                 _optp(logs and logs["tbs_bits"], C.c_int))
        if rc:
            raise RuntimeError(f"rso_run_trace rc={rc}")
        return logs

    def run_synth(self, cqi_epochs, seed, n_ttis, refresh=40, phy_error_draws=0, log=True, per_prb=False):
        """per_prb=True: cqi_epochs [n_epochs][U][R*rbg_size]."""
        e = np.ascontiguousarray(cqi_epochs, np.uint8)
        assert e.shape[1:] == (self.U, self.R * (self.rbg_size if per_prb else 1))
        logs = None
        if log:
            logs = {"rbg_to_user": np.zeros((n_ttis, self.R), np.int32),
                    "tbs_bits": np.zeros((n_ttis, self.U), np.int32)}
        fn = lib().rso_run_synth_prb if per_prb else lib().rso_run_synth
        rc = fn(self.h, _p(e, C.c_uint8), e.shape[0], refresh, seed, phy_error_draws,
                n_ttis, _optp(logs and logs["rbg_to_user"], C.c_int),
                _optp(logs and logs["tbs_bits"], C.c_int))
        if rc:
            raise RuntimeError(f"rso_run_synth rc={rc}")
        return logs


def run_synth_many(template, n_cells, cqi_epochs, seeds, n_ttis, threads=0, refresh=40, phy_error_draws=0):
    """n_cells independent cells configured like `template` (a Cell), OpenMP over the host cores.
    Returns (total bytes granted, OpenMP team size)."""
    e = np.ascontiguousarray(cqi_epochs, np.uint8)
    assert e.shape[1:] == (template.U, template.R)
    sd = np.ascontiguousarray(seeds, np.uint32)
    assert sd.shape == (n_cells,)
    total, used = C.c_int64(0), C.c_int(0)
    rc = lib().rso_run_synth_many(C.byref(template.cfg), n_cells, _p(e, C.c_uint8), e.shape[0], refresh, _p(sd, C.c_uint),
                                  phy_error_draws, n_ttis, threads, C.byref(total), C.byref(used))
    if rc:
        raise RuntimeError(f"rso_run_synth_many rc={rc}")
    return int(total.value), int(used.value)


def run_synth_cells(template, cqi_epochs, seeds, n_ttis, threads=0, refresh=40, phy_error_draws=0):
    """Independent cells configured like `template`, each on its own epochs cqi_epochs [n_cells][n_epochs][U][R], OpenMP over the host
    cores.  Returns the cells' final state: dict(avg_rate, cum_bytes, cum_rbs [n_cells][U], slice_state [n_cells][S], threads)."""
    e = np.ascontiguousarray(cqi_epochs, np.uint8)
    n_cells = e.shape[0]
    assert e.ndim == 4 and e.shape[2:] == (template.U, template.R)
    sd = np.ascontiguousarray(seeds, np.uint32)
    assert sd.shape == (n_cells,)
    avg = np.zeros((n_cells, template.U), np.float64)
    cb = np.zeros((n_cells, template.U), np.int64)
    cr = np.zeros((n_cells, template.U), np.int64)
    sl = np.zeros((n_cells, template.S), np.float64)
    used = C.c_int(0)
    rc = lib().rso_run_synth_cells(C.byref(template.cfg), n_cells, _p(e, C.c_uint8), e.shape[1], refresh, _p(sd, C.c_uint), phy_error_draws,
                                   n_ttis, threads, _p(avg, C.c_double), _p(cb, C.c_int64), _p(cr, C.c_int64), _p(sl, C.c_double), C.byref(used))
    if rc:
        raise RuntimeError(f"rso_run_synth_cells rc={rc}")
    return {"avg_rate": avg, "cum_bytes": cb, "cum_rbs": cr, "slice_state": sl, "threads": int(used.value)}


def ref_lib(name):
    """Load a prebuilt reference piece from oracle/_ref (None if absent)."""
    p = REF_DIR / name
    if not p.exists():
        return None
    return C.CDLL(str(p))
