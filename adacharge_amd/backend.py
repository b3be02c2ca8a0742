"""ctypes binding of the C ABI in include/acn_qp.h (libacn_qp_hip.so).

There is exactly one compute path: the HIP library.  If it is missing or no GPU
is visible, everything here raises -- there is no CPU fallback by design
(the CPU implementations under oracle/ are test infrastructure and are never
imported from this package).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .builder import CONE_SOC, ProblemBatch, SiteData

_LIB_NAME = "libacn_qp_hip.so"
_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")

STATUS_UNSET = 0
STATUS_SOLVED = 1
STATUS_MAX_ITER = 2
STATUS_PRIMAL_INFEASIBLE = 3
STATUS_EMPTY_SET = 4
STATUS_SOLVED_INACCURATE = 5   # cp.OPTIMAL_INACCURATE: accepted by the reference (aco.py:319)
ACCEPTED_STATUSES = (STATUS_SOLVED, STATUS_SOLVED_INACCURATE)
STATUS_NAMES = {
    STATUS_UNSET: "unset",
    STATUS_SOLVED: "optimal",
    STATUS_MAX_ITER: "max_iter_reached",
    STATUS_PRIMAL_INFEASIBLE: "infeasible",
    STATUS_EMPTY_SET: "infeasible",
    STATUS_SOLVED_INACCURATE: "optimal_inaccurate",
}


class BackendUnavailable(RuntimeError):
    """The HIP library could not be loaded or no MI355X is visible."""


class _Site(C.Structure):
    _fields_ = [
        ("n_evse", C.c_int32),
        ("n_infra", C.c_int32),
        ("n_rows", C.c_int32),
        ("cone", C.c_int32),
        ("has_peak", C.c_int32),
        ("has_flat", C.c_int32),
        ("has_max", C.c_int32),
        ("G", C.c_void_p),
        ("limits", C.c_void_p),
    ]


class _Problems(C.Structure):
    _fields_ = [
        ("batch", C.c_int32),
        ("t_max", C.c_int32),
        ("k_sessions", C.c_int32),
        ("horizon", C.c_void_p),
        ("lb", C.c_void_p),
        ("ub", C.c_void_p),
        ("q", C.c_void_p),
        ("pdiag", C.c_void_p),
        ("s_off", C.c_void_p),
        ("s_len", C.c_void_p),
        ("s_cap", C.c_void_p),
        ("s_eq", C.c_void_p),
        ("peak", C.c_void_p),
        ("lf", C.c_void_p),
        ("dc", C.c_void_p),
        ("dfloor", C.c_void_p),
        ("warm_x", C.c_void_p),
        ("warm_y", C.c_void_p),
    ]


class _Table(C.Structure):
    """Mirror of ``acnqp_table`` (include/acn_qp.h)."""

    _fields_ = [(k, C.c_int32) for k in ("batch", "t_max", "k_sessions", "n_sessions", "n_horizons")] + [
        (k, C.c_void_p) for k in ("horizon", "q_index", "q_table", "pdiag", "s_eq", "peak", "lf", "dc", "dfloor", "sess_seg",
                                  "s_evse", "s_slot", "s_off", "s_len", "s_cap", "rate_seg", "min_rates", "max_rates")]


class _Results(C.Structure):
    _fields_ = [
        ("x", C.c_void_p),
        ("status", C.c_void_p),
        ("iters", C.c_void_p),
        ("pri_res", C.c_void_p),
        ("dua_res", C.c_void_p),
        ("obj", C.c_void_p),
        ("y", C.c_void_p),
        ("x_dev", C.c_void_p),
    ]


class Options(C.Structure):
    """Mirror of ``acnqp_options``; construct with ``default_options()``."""

    _fields_ = [
        ("eps_abs", C.c_double),
        ("eps_rel", C.c_double),
        ("max_iter", C.c_int32),
        ("check_every", C.c_int32),
        ("adapt_every", C.c_int32),
        ("rho", C.c_double),
        ("sigma", C.c_double),
        ("alpha", C.c_double),
        ("adapt_tol", C.c_double),
        ("reg_rel", C.c_double),
        ("precision", C.c_int32),
        ("accel_mem", C.c_int32),
        ("stall_iters", C.c_int32),
        ("retry_passes", C.c_int32),
        ("retry_max_iter", C.c_int32),
        ("polish_iters", C.c_int32),
        ("retry_rho", C.c_double),
        ("inaccurate_floor", C.c_double),
        ("polish_stall", C.c_int32),
    ]


# every symbol include/acn_qp.h declares; tests check the library exports all of them
EXPORTED_SYMBOLS = (
    "acnqp_create",
    "acnqp_solve_batch",
    "acnqp_solve_batch_device",
    "acnqp_destroy",
    "acnqp_default_options",
    "acnqp_last_error",
    "acnqp_abi_version",
    "acnqp_last_kernel_ms",
    "acnqp_accel_columns",
    "acnqp_kernel_times",
    "acnqp_ordered_launch_count",
    "acnqp_polish_stats",
    "acnqp_solve_batches",
    "acnqp_solve_table",
    "acnqp_host_alloc",
    "acnqp_host_free",
    "acnqp_launch_count",
)

_lib = None


def library_path() -> str:
    return os.environ.get("ACNQP_LIBRARY", os.path.join(_LIB_DIR, _LIB_NAME))


def load_library():
    """dlopen the HIP library (once) and set the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime; it must be the one already loaded when our
    # library resolves libamdhip64, or the process ends up with two runtimes and the
    # second sees no GPU.  torch is plumbing here (device memory, torch.distributed).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise BackendUnavailable(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  adacharge_amd has no CPU fallback."
        )
    try:
        lib = C.CDLL(path)
    except OSError as exc:  # missing ROCm runtime etc.
        raise BackendUnavailable(f"cannot load {path}: {exc}") from exc
    lib.acnqp_create.argtypes = [C.POINTER(_Site), C.c_int32, C.POINTER(C.c_void_p)]
    lib.acnqp_create.restype = C.c_int
    lib.acnqp_solve_batch.argtypes = [C.c_void_p, C.POINTER(_Problems), C.POINTER(Options), C.POINTER(_Results)]
    lib.acnqp_solve_batch.restype = C.c_int
    lib.acnqp_solve_batch_device.argtypes = [
        C.c_void_p, C.POINTER(_Problems), C.POINTER(Options), C.POINTER(_Results), C.c_void_p,
    ]
    lib.acnqp_solve_batch_device.restype = C.c_int
    lib.acnqp_destroy.argtypes = [C.c_void_p]
    lib.acnqp_destroy.restype = None
    lib.acnqp_default_options.argtypes = [C.POINTER(Options)]
    lib.acnqp_default_options.restype = None
    lib.acnqp_last_error.argtypes = []
    lib.acnqp_last_error.restype = C.c_char_p
    lib.acnqp_abi_version.argtypes = []
    lib.acnqp_abi_version.restype = C.c_int32
    lib.acnqp_last_kernel_ms.argtypes = [C.c_void_p]
    lib.acnqp_last_kernel_ms.restype = C.c_float
    lib.acnqp_accel_columns.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    lib.acnqp_accel_columns.restype = C.c_int32
    lib.acnqp_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32]
    lib.acnqp_kernel_times.restype = C.c_int32
    lib.acnqp_solve_batches.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_Problems), C.POINTER(Options), C.POINTER(_Results)]
    lib.acnqp_solve_batches.restype = C.c_int
    lib.acnqp_solve_table.argtypes = [C.c_void_p, C.POINTER(_Table), C.POINTER(Options), C.POINTER(_Results)]
    lib.acnqp_solve_table.restype = C.c_int
    lib.acnqp_host_alloc.argtypes = [C.c_size_t]
    lib.acnqp_host_alloc.restype = C.c_void_p
    lib.acnqp_host_free.argtypes = [C.c_void_p]
    lib.acnqp_host_free.restype = None
    lib.acnqp_launch_count.argtypes = [C.c_void_p]
    lib.acnqp_launch_count.restype = C.c_int64
    lib.acnqp_ordered_launch_count.argtypes = [C.c_void_p]
    lib.acnqp_ordered_launch_count.restype = C.c_int64
    lib.acnqp_polish_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    lib.acnqp_polish_stats.restype = C.c_int
    _lib = lib
    return lib


def default_options(**overrides) -> Options:
    o = Options()
    load_library().acnqp_default_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown solver option {k!r}")
        setattr(o, k, v)
    return o


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().acnqp_last_error().decode()
        if rc == -3:
            raise BackendUnavailable(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class BatchResult:
    x: np.ndarray        # (B, N, Tm)
    status: np.ndarray   # (B,) int32
    iters: np.ndarray    # (B,) int32
    pri_res: np.ndarray
    dua_res: np.ndarray
    obj: np.ndarray
    kernel_ms: float = float("nan")   # sum of the HIP-event durations of the call's launches (chunks of the pipelined
                                      # entry overlap on the GPU: an upper bound of the time the GPU was busy)
    y: Optional[np.ndarray] = None   # (B, Mg, Tm) multipliers of the site rows (when asked for): warm_y of a later solve


class _PinnedBlock:
    """Owner of one acnqp_host_alloc block; freed when the last numpy view of it is gone."""

    def __init__(self, nbytes: int):
        self._lib = load_library()
        self.nbytes = int(nbytes)
        self.ptr = self._lib.acnqp_host_alloc(self.nbytes)
        if not self.ptr:
            raise MemoryError(f"acnqp_host_alloc({nbytes}) failed: {self._lib.acnqp_last_error().decode()}")

    def __del__(self):
        try:
            if self.ptr:
                self._lib.acnqp_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """A zero-filled numpy array in pinned host memory (acnqp_host_alloc): the library's H2D / D2H copies of such
    arrays are direct DMA.  Same call shape as ``np.zeros``; use it as ``alloc=`` of ``builder.build_batch``."""
    dt = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(k) for k in shape)
    n = int(np.prod(shape)) if shape else 1
    block = _PinnedBlock(max(n * dt.itemsize, 1))
    buf = (C.c_char * block.nbytes).from_address(block.ptr)
    buf._owner = block   # numpy keeps `buf` alive through .base; `buf` keeps the block
    a = np.frombuffer(buf, dtype=dt, count=n).reshape(shape)
    a[...] = 0
    return a


class SiteHandle:
    """One ``acnqp_handle``: a site uploaded to one GPU."""

    def __init__(self, site: SiteData, device: int = 0):
        self._lib = load_library()
        self.site = site
        self.device = int(device)
        G = np.ascontiguousarray(site.G, dtype=np.float64)
        lim = np.ascontiguousarray(site.limits, dtype=np.float64)
        desc = _Site(
            site.N, site.M, site.Mg, 1 if site.cone == CONE_SOC else 0, 1 if site.has_peak else 0,
            1 if site.has_flat else 0, 1 if site.has_max else 0,
            _ptr(G) if G.size else None, _ptr(lim) if lim.size else None,
        )
        h = C.c_void_p()
        _check(self._lib.acnqp_create(C.byref(desc), self.device, C.byref(h)), "acnqp_create")
        self._h = h
        self._launches_seen = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.acnqp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- host buffers -----------------------------------------------------------
    def _check_site(self, batch: ProblemBatch):
        if batch.site is not self.site and (
            batch.site.Mg != self.site.Mg or batch.site.N != self.site.N or batch.site.cone != self.site.cone
            or batch.site.has_flat != self.site.has_flat or batch.site.has_max != self.site.has_max
        ):
            raise ValueError("batch was built for a different site than this handle")

    def _marshal(self, batch: ProblemBatch, pinned: bool, x_dev=None, warm=None, want_y=False):
        """ctypes views of one batch: (_Problems, _Results, BatchResult, keep-alive list).  Arrays that already
        are C-contiguous with the ABI's dtype are passed as they are (e.g. pinned arrays from ``pinned_empty``)."""
        B, N, Tm = batch.B, batch.N, batch.Tm
        arrs = dict(
            horizon=np.ascontiguousarray(batch.T, np.int32),
            lb=np.ascontiguousarray(batch.lb, np.float64),
            ub=np.ascontiguousarray(batch.ub, np.float64),
            q=np.ascontiguousarray(batch.q, np.float64),
            pdiag=np.ascontiguousarray(batch.pdiag, np.float64),
            s_off=np.ascontiguousarray(batch.s_off, np.int32),
            s_len=np.ascontiguousarray(batch.s_len, np.int32),
            s_cap=np.ascontiguousarray(batch.s_cap, np.float64),
            s_eq=np.ascontiguousarray(batch.s_eq, np.uint8),
        )
        peak = None if batch.peak is None else np.ascontiguousarray(batch.peak, np.float64)
        lf = np.ascontiguousarray(batch.lf, np.float64) if self.site.has_flat else None
        dc = np.ascontiguousarray(batch.dc, np.float64) if self.site.has_max else None
        dfl = np.ascontiguousarray(batch.dfloor, np.float64) if self.site.has_max else None
        wx = wy = None
        if warm is not None:
            wx = np.ascontiguousarray(warm[0], np.float64)
            wy = np.ascontiguousarray(warm[1], np.float64)
            if wx.shape != (B, N, Tm) or wy.shape != (B, self.site.Mg, Tm):
                raise ValueError(f"warm start arrays must have shapes {(B, N, Tm)} and {(B, self.site.Mg, Tm)}")
        p = _Problems(B, Tm, batch.K, *[_ptr(arrs[k]) for k in
                                       ("horizon", "lb", "ub", "q", "pdiag", "s_off", "s_len", "s_cap", "s_eq")],
                      _ptr(peak), _ptr(lf), _ptr(dc), _ptr(dfl), _ptr(wx), _ptr(wy))
        new = pinned_empty if pinned else (lambda shape, dtype=np.float64: np.zeros(shape, dtype))
        res = BatchResult(
            new((B, N, Tm)), new(B, np.int32), new(B, np.int32), new(B), new(B), new(B),
        )
        if want_y:
            res.y = new((B, self.site.Mg, Tm))
        r = _Results(_ptr(res.x), _ptr(res.status), _ptr(res.iters), _ptr(res.pri_res), _ptr(res.dua_res), _ptr(res.obj),
                     _ptr(res.y), None if x_dev is None else C.c_void_p(int(x_dev)))
        return p, r, res, (arrs, peak, lf, dc, dfl, wx, wy)

    def _finish(self, batch: ProblemBatch, res: "BatchResult"):
        if self.site.has_flat:   # the kernel's obj covers pdiag and q; add 1/2 lf sum_t (v' x_t)^2
            v = self.site.G[self.site.flat_row]
            res.obj = res.obj + 0.5 * batch.lf * np.einsum("n,bnt->bt", v, res.x).__pow__(2).sum(axis=1)
        if self.site.has_max:   # ... and dc * max(max_t v' x_t, dfloor)
            v = self.site.G[self.site.max_row]
            agg = np.einsum("n,bnt->bt", v, res.x)
            res.obj = res.obj + batch.dc * np.maximum(agg.max(axis=1), batch.dfloor)
        if batch.presolve_status is not None:
            res.status[batch.presolve_status != 0] = STATUS_EMPTY_SET
        return res

    def solve(self, batch: ProblemBatch, options: Optional[Options] = None, pinned_results: bool = False,
              warm=None, want_y: bool = False) -> BatchResult:
        """acnqp_solve_batch: one batch, host buffers in and out, synchronous (pipelined in chunks inside).
        ``warm = (x0, y0)``: optional warm start (an earlier schedule (B, N, Tm) and its ``BatchResult.y`` (B, Mg, Tm),
        shifted by the caller); ``want_y``: also return the site-row multipliers ``y`` for a later warm start.
        Problems a pass leaves SOLVED_INACCURATE / MAX_ITER on a plateau are re-solved inside the library
        (``options.retry_passes``, include/acn_qp.h) -- every entry point behaves the same."""
        self._check_site(batch)
        o = options if options is not None else default_options()
        p, r, res, keep = self._marshal(batch, pinned_results, warm=warm, want_y=want_y)
        self.kernel_times()   # forget earlier launches: kernel_ms below is the sum over THIS call's launches (chunks)
        self._launches_seen = int(self._lib.acnqp_launch_count(self._h))
        _check(self._lib.acnqp_solve_batch(self._h, C.byref(p), C.byref(o), C.byref(r)), "acnqp_solve_batch")
        del keep
        res.kernel_ms = self._kernel_ms_of_call()
        return self._finish(batch, res)

    def solve_table(self, plan, options: Optional[Options] = None, pinned_results: bool = False, want_y: bool = False,
                    out: Optional["BatchResult"] = None) -> "BatchResult":
        """acnqp_solve_table: a ``builder.TablePlan`` (sessions + one linear cost per horizon) in, schedules out; the
        dense (B, N, Tm) problem arrays are formed on the device.  Same results as ``solve(plan.expand())``.
        ``out``: a BatchResult of an earlier call of the same shape to write into (a service that solves every control
        period keeps its -- pinned -- result arrays instead of allocating 85 MB per call)."""
        if plan.site is not self.site and (plan.site.N, plan.site.Mg, plan.site.cone) != (self.site.N, self.site.Mg, self.site.cone):
            raise ValueError("plan was built for another site")
        o = options if options is not None else default_options()
        B, N, Tm = plan.B, plan.N, plan.Tm
        c = lambda a, dt: None if a is None else np.ascontiguousarray(a, dt)
        keep = dict(
            horizon=c(plan.T, np.int32), q_index=c(plan.q_index, np.int32), q_table=c(plan.q_table, np.float64),
            pdiag=c(plan.pdiag, np.float64), s_eq=c(plan.s_eq, np.uint8), peak=c(plan.peak, np.float64) if self.site.has_peak else None,
            lf=c(plan.lf, np.float64) if self.site.has_flat else None, dc=c(plan.dc, np.float64) if self.site.has_max else None,
            dfloor=c(plan.dfloor, np.float64) if self.site.has_max else None, sess_seg=c(plan.sess_seg, np.int32),
            s_evse=c(plan.s_evse, np.int32), s_slot=c(plan.s_slot, np.int32), s_off=c(plan.s_off, np.int32), s_len=c(plan.s_len, np.int32),
            s_cap=c(plan.s_cap, np.float64), rate_seg=c(plan.rate_seg, np.int32), min_rates=c(plan.min_rates, np.float64),
            max_rates=c(plan.max_rates, np.float64))
        t = _Table(B, Tm, plan.K, plan.S, len(plan.q_table), *[_ptr(keep[k]) for k in (
            "horizon", "q_index", "q_table", "pdiag", "s_eq", "peak", "lf", "dc", "dfloor", "sess_seg", "s_evse", "s_slot", "s_off",
            "s_len", "s_cap", "rate_seg", "min_rates", "max_rates")])
        new = pinned_empty if pinned_results else (lambda shape, dtype=np.float64: np.zeros(shape, dtype))
        if out is not None and out.x.shape == (B, N, Tm) and (not want_y or (out.y is not None and out.y.shape == (B, self.site.Mg, Tm))):
            res = out
        else:
            res = BatchResult(new((B, N, Tm)), new(B, np.int32), new(B, np.int32), new(B), new(B), new(B))
            if want_y:
                res.y = new((B, self.site.Mg, Tm))
        r = _Results(_ptr(res.x), _ptr(res.status), _ptr(res.iters), _ptr(res.pri_res), _ptr(res.dua_res), _ptr(res.obj),
                     _ptr(res.y) if want_y else None, None)
        self.kernel_times()
        self._launches_seen = int(self._lib.acnqp_launch_count(self._h))
        _check(self._lib.acnqp_solve_table(self._h, C.byref(t), C.byref(o), C.byref(r)), "acnqp_solve_table")
        del keep
        res.kernel_ms = self._kernel_ms_of_call()
        return self._finish(plan, res)

    def solve_many(self, batches, options: Optional[Options] = None, pinned_results: bool = True):
        """acnqp_solve_batches: several independent batches in ONE pipelined pass (shared launches, overlapped
        copies).  Returns one BatchResult per batch."""
        o = options if options is not None else default_options()
        n = len(batches)
        P, R = (_Problems * n)(), (_Results * n)()
        results, keep = [], []
        for g, batch in enumerate(batches):
            self._check_site(batch)
            P[g], R[g], res, k = self._marshal(batch, pinned_results)
            results.append(res)
            keep.append(k)
        _check(self._lib.acnqp_solve_batches(self._h, n, P, C.byref(o), R), "acnqp_solve_batches")
        del keep
        return [self._finish(b, r) for b, r in zip(batches, results)]

    def prepare_many(self, batches, pinned_results: bool = True, x_dev_ptrs=None):
        """Marshal once, solve repeatedly (bench.py): returns a callable ``run(options)`` that re-submits the same
        host buffers through acnqp_solve_batches and the list of BatchResults it fills.  ``x_dev_ptrs[g]``: optional
        device address that also receives batch g's schedules (acnqp_results.x_dev)."""
        n = len(batches)
        P, R = (_Problems * n)(), (_Results * n)()
        results, keep = [], []
        for g, batch in enumerate(batches):
            self._check_site(batch)
            P[g], R[g], res, k = self._marshal(batch, pinned_results, None if x_dev_ptrs is None else x_dev_ptrs[g])
            results.append(res)
            keep.append(k)

        def run(options: Options):
            _check(self._lib.acnqp_solve_batches(self._h, n, P, C.byref(options), R), "acnqp_solve_batches")
            return results

        run.keep = keep
        return run, results

    # -- device buffers (torch tensors or any object with data_ptr()) --------------
    def solve_device(self, dev: "DeviceBatch", options: Optional[Options] = None, stream: int = 0) -> None:
        o = options if options is not None else default_options()
        p = _Problems(
            dev.B, dev.Tm, dev.K,
            dev.horizon.data_ptr(), dev.lb.data_ptr(), dev.ub.data_ptr(), dev.q.data_ptr(), dev.pdiag.data_ptr(),
            dev.s_off.data_ptr(), dev.s_len.data_ptr(), dev.s_cap.data_ptr(), dev.s_eq.data_ptr(),
            None if dev.peak is None else dev.peak.data_ptr(),
            dev.lf.data_ptr() if self.site.has_flat else None,
            dev.dc.data_ptr() if self.site.has_max else None,
            dev.dfloor.data_ptr() if self.site.has_max else None,
            None, None,
        )
        r = _Results(dev.x.data_ptr(), dev.status.data_ptr(), dev.iters.data_ptr(),
                     dev.pri_res.data_ptr(), dev.dua_res.data_ptr(), dev.obj.data_ptr(), None, None)
        _check(
            self._lib.acnqp_solve_batch_device(self._h, C.byref(p), C.byref(o), C.byref(r), C.c_void_p(stream)),
            "acnqp_solve_batch_device",
        )

    def _kernel_ms_of_call(self) -> float:
        """Sum of the HIP-event durations of the launches since the previous ``kernel_times`` call; NaN when an event
        could not be read or when the call made more launches than the library's 64-entry event ring holds (the sum
        would silently under-report)."""
        before = int(self._lib.acnqp_launch_count(self._h))
        ms = self.kernel_times()
        if any(m < 0 for m in ms) or before - self._launches_seen > len(ms):
            self._launches_seen = before
            return float("nan")
        self._launches_seen = before
        return float(sum(ms))

    def polish_stats(self) -> dict:
        """Counters of the device-side polish over this handle's life (acnqp_polish_stats; synchronises)."""
        buf = (C.c_int64 * 16)()
        _check(self._lib.acnqp_polish_stats(self._h, buf, 16), "acnqp_polish_stats")
        out = dict(zip(("attempted", "solved", "gave_up_rows", "gave_up_pivot", "gave_up_rounds", "gave_up_kkt", "rounds"), [int(v) for v in buf[:7]]))
        out["phase_us"] = [int(v) // 100 for v in buf[8:16]]   # summed over the polish workgroups
        return out

    def ordered_launches(self) -> int:
        """Launches of this handle whose queue order was sorted by session count (acnqp_ordered_launch_count)."""
        return int(self._lib.acnqp_ordered_launch_count(self._h))

    def last_kernel_ms(self) -> float:
        return float(self._lib.acnqp_last_kernel_ms(self._h))

    def kernel_times(self, capacity: int = 64):
        """Durations (ms) of the launches since the previous call (at most the 64 most recent); blocks
        until they have finished."""
        buf = (C.c_float * int(capacity))()
        n = int(self._lib.acnqp_kernel_times(self._h, buf, int(capacity)))
        return [float(buf[k]) for k in range(n)]

    def accel_columns(self, t_max: int, k_sessions: int, options: Options) -> int:
        """Anderson columns the kernels use for this problem shape under ``options`` (shape-only rule)."""
        return int(self._lib.acnqp_accel_columns(self._h, int(t_max), int(k_sessions), int(options.precision),
                                                 int(options.accel_mem)))


class DeviceBatch:
    """A ProblemBatch resident in HBM (torch tensors on one GPU) plus result
    tensors; torch is used for device memory only."""

    def __init__(self, batch: ProblemBatch, device):
        import torch

        dev = torch.device(device)
        self.B, self.N, self.Tm, self.K = batch.B, batch.N, batch.Tm, batch.K
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
        self.horizon = t(batch.T, np.int32)
        self.lb = t(batch.lb, np.float64)
        self.ub = t(batch.ub, np.float64)
        self.q = t(batch.q, np.float64)
        self.pdiag = t(batch.pdiag, np.float64)
        self.s_off = t(batch.s_off, np.int32)
        self.s_len = t(batch.s_len, np.int32)
        self.s_cap = t(batch.s_cap, np.float64)
        self.s_eq = t(batch.s_eq, np.uint8)
        self.peak = None if batch.peak is None else t(batch.peak, np.float64)
        self.lf = t(batch.lf, np.float64)
        self.dc = t(batch.dc if batch.dc is not None else np.zeros(batch.B), np.float64)
        self.dfloor = t(batch.dfloor if batch.dfloor is not None else np.zeros(batch.B), np.float64)
        self.x = torch.zeros((self.B, self.N, self.Tm), dtype=torch.float64, device=dev)
        self.status = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.iters = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.pri_res = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.dua_res = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.obj = torch.zeros(self.B, dtype=torch.float64, device=dev)
