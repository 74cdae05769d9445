"""ctypes binding of libmmgp.so (include/mmgp.h).

Plumbing only: every class below is a thin handle wrapper whose methods call one
C-ABI entry point.  There is no Python/NumPy compute path: if the shared library
is missing, or no HIP device is usable, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.environ.get("MMGP_LIBDIR") or os.path.dirname(os.path.abspath(__file__))  # MMGP_LIBDIR: A/B builds
LIB_PATH = os.path.join(_PKG, "libmmgp.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)


class LevelDesc(C.Structure):
    _fields_ = [("n", C.c_int), ("a_size", C.c_int), ("rowptr", _ip), ("col", _ip), ("val", _dp),
                ("bcflags", _ip), ("neumann_flag", C.c_int), ("omega", C.c_double), ("iters", C.c_int),
                ("nb", C.c_int), ("btype", _ip), ("bptr", _ip), ("bpts", _ip), ("bvals", _dp),
                ("tile_ptr", _ip), ("n_tiles", C.c_int), ("tile_size", C.c_int), ("lanes_per_row", C.c_int),
                ("tile_phase", _ip), ("waves_per_tile", C.c_int)]


class LevelInfo(C.Structure):
    _fields_ = [("n_tiles", C.c_int), ("n_phases", C.c_int), ("n_groups", C.c_int), ("lanes_per_row", C.c_int),
                ("max_lds_bytes", C.c_int), ("sor_rows", C.c_longlong), ("sor_nnz", C.c_longlong),
                ("stream_bytes", C.c_longlong), ("halo_entries", C.c_longlong), ("neumann_rows", C.c_longlong),
                ("waves_per_tile", C.c_int), ("max_tile_levels", C.c_int)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


#: every symbol include/mmgp.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "mmg_last_error", "mmg_device_count", "mmg_set_device", "mmg_set_stream", "mmg_synchronize",
    "mmg_device_props", "mmg_auto_tile_points", "mmg_set_option", "mmg_get_counter", "mmg_comm_get_unique_id", "mmg_comm_init", "mmg_comm_finalize",
    "mmg_level_set_exchange", "mmg_level_set_exchange_mode", "mmg_level_point_phases", "mmg_level_exchange",
    "mmg_level_create", "mmg_level_destroy", "mmg_level_info_get", "mmg_level_set_x", "mmg_level_get_x",
    "mmg_level_set_rhs", "mmg_level_get_rhs", "mmg_level_set_bvals", "mmg_level_set_omega_iters",
    "mmg_level_sor", "mmg_level_sweeps", "mmg_level_bound_eval_neumann", "mmg_level_residual",
    "mmg_level_residual_ratio", "mmg_level_boundary_op", "mmg_level_modify_coeff_neumann", "mmg_level_zero_x",
    "mmg_level_time_sweeps", "mmg_level_time_residual", "mmg_level_time_phases", "mmg_transfer_create", "mmg_transfer_destroy",
    "mmg_restrict", "mmg_prolong_add", "mmg_hierarchy_create", "mmg_hierarchy_destroy", "mmg_vcycle",
    "mmg_hierarchy_residual", "mmg_vcycles", "mmg_rbf_weights", "mmg_spmv_create", "mmg_spmv_destroy", "mmg_spmv_apply", "mmg_fracstep_create", "mmg_fracstep_destroy",
    "mmg_fracstep_set", "mmg_fracstep_get", "mmg_fracstep_calc_hat", "mmg_fracstep_set_ppe_source",
    "mmg_fracstep_correct", "mmg_fracstep_residual", "mmg_fracstep_create_3d", "mmg_fracstep_set_bound_values",
    "mmg_fracstep_apply_bound", "mmg_fracstep_step", "mmg_level_set_neumann_coupling", "mmg_level_push_inhomog_to_rhs",
    "mmg_hierarchy_set_gather",
    "mmg_knn",
    "mmg_hierarchy_set_correction_damping",
    "mmg_host_threads",
    "mmg_rbf_stencils",
    "mmg_comm_info", "mmg_level_exchange_info", "mmg_level_time_exchange",
]

_lib = None


class MmgError(RuntimeError):
    pass


def lib():
    """Load libmmgp.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MmgError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C meshlessmultigridpoisson_amd/csrc`")
        L = C.CDLL(LIB_PATH)
        L.mmg_last_error.restype = C.c_char_p
        vp = C.c_void_p
        L.mmg_level_create.argtypes = [C.POINTER(vp), C.POINTER(LevelDesc)]
        L.mmg_level_destroy.argtypes = [vp]
        L.mmg_level_destroy.restype = None
        L.mmg_level_info_get.argtypes = [vp, C.POINTER(LevelInfo)]
        for f in ("mmg_level_set_x", "mmg_level_get_x", "mmg_level_set_rhs", "mmg_level_get_rhs",
                  "mmg_level_set_bvals", "mmg_level_residual"):
            getattr(L, f).argtypes = [vp, _dp, C.c_int]
        L.mmg_level_set_omega_iters.argtypes = [vp, C.c_double, C.c_int]
        for f in ("mmg_level_sor", "mmg_level_bound_eval_neumann", "mmg_level_zero_x"):
            getattr(L, f).argtypes = [vp]
        for f in ("mmg_level_sweeps", "mmg_level_boundary_op", "mmg_level_modify_coeff_neumann"):
            getattr(L, f).argtypes = [vp, C.c_int]
        L.mmg_level_residual_ratio.argtypes = [vp, _dp]
        L.mmg_level_time_sweeps.argtypes = [vp, C.c_int, C.c_int, _fp]
        L.mmg_level_time_residual.argtypes = [vp, C.c_int, _fp]
        L.mmg_level_time_phases.argtypes = [vp, C.c_int, _fp, _ip]
        L.mmg_level_time_exchange.argtypes = [vp, C.c_int, _fp]
        L.mmg_level_exchange_info.argtypes = [vp, _ip, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.mmg_comm_info.argtypes = [_ip, _ip]
        L.mmg_transfer_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _ip, _ip, _dp, C.c_int]
        L.mmg_transfer_destroy.argtypes = [vp]
        L.mmg_transfer_destroy.restype = None
        L.mmg_restrict.argtypes = [vp, vp, vp]
        L.mmg_prolong_add.argtypes = [vp, vp, vp]
        L.mmg_hierarchy_create.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int]
        L.mmg_hierarchy_destroy.argtypes = [vp]
        L.mmg_hierarchy_destroy.restype = None
        L.mmg_vcycle.argtypes = [vp, _dp]
        L.mmg_hierarchy_residual.argtypes = [vp, _dp]
        L.mmg_vcycles.argtypes = [vp, C.c_int, _dp, _fp]
        L.mmg_spmv_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _ip, _ip, _dp]
        L.mmg_spmv_destroy.argtypes = [vp]
        L.mmg_spmv_destroy.restype = None
        L.mmg_spmv_apply.argtypes = [vp, _dp, C.c_int, _dp, C.c_int]
        L.mmg_fracstep_create.argtypes = [C.POINTER(vp), vp, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp, _dp, _dp,
                                          _ip, C.c_int]
        L.mmg_fracstep_destroy.argtypes = [vp]
        L.mmg_fracstep_destroy.restype = None
        L.mmg_fracstep_set.argtypes = [vp, C.c_int, _dp, C.c_int]
        L.mmg_fracstep_get.argtypes = [vp, C.c_int, _dp, C.c_int]
        L.mmg_fracstep_calc_hat.argtypes = [vp, C.c_double, C.c_double, C.c_double]
        L.mmg_fracstep_set_ppe_source.argtypes = [vp, C.c_double, C.c_double]
        L.mmg_fracstep_correct.argtypes = [vp, C.c_double, C.c_double]
        L.mmg_fracstep_residual.argtypes = [vp, _dp]
        L.mmg_set_stream.argtypes = [vp]
        L.mmg_set_device.argtypes = [C.c_int]
        L.mmg_device_count.argtypes = [_ip]
        L.mmg_device_props.argtypes = [_ip, _ip]
        L.mmg_set_option.argtypes = [C.c_char_p, C.c_int]
        L.mmg_comm_get_unique_id.argtypes = [C.c_char_p]
        L.mmg_comm_init.argtypes = [C.c_int, C.c_int, C.c_char_p]
        L.mmg_level_set_exchange.argtypes = [vp, C.c_int, C.c_int, _ip, _ip, _ip, _ip]
        L.mmg_level_exchange.argtypes = [vp]
        L.mmg_auto_tile_points.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise MmgError(f"libmmgp error {rc}: {lib().mmg_last_error().decode()}")


def auto_tile_points(n_points, dim, stencil, lanes_per_row=0, compute_units=0, lds_bytes_per_cu=0):
    return lib().mmg_auto_tile_points(int(n_points), int(dim), int(stencil), int(lanes_per_row), int(compute_units),
                                      int(lds_bytes_per_cu))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    check(lib().mmg_comm_get_unique_id(buf))
    return buf.raw


def comm_init(rank, nranks, id128):
    check(lib().mmg_comm_init(int(rank), int(nranks), id128))


def comm_finalize():
    check(lib().mmg_comm_finalize())


def comm_info():
    """(ranks, rank) read back from RCCL (ncclCommCount / ncclCommUserRank)"""
    n, r = C.c_int(0), C.c_int(-1)
    check(lib().mmg_comm_info(C.byref(n), C.byref(r)))
    return n.value, r.value


def set_option(name, value):
    check(lib().mmg_set_option(name.encode(), int(value)))


def get_counter(name):
    v = C.c_longlong(0)
    f = lib().mmg_get_counter
    f.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
    check(f(name.encode(), C.byref(v)))
    return v.value


def device_props():
    cu, lds = C.c_int(0), C.c_int(0)
    check(lib().mmg_device_props(C.byref(cu), C.byref(lds)))
    return cu.value, lds.value


def rbf_weights(dim, polydeg, rbf_exp, cloud_xyz, eval_xyz, nbr, ops):
    """mmg_rbf_weights: batched RBF-FD stencil weights; returns [n_ops][n_eval][stencil]."""
    cloud = np.ascontiguousarray(cloud_xyz, dtype=np.float64).reshape(-1, 3)
    ev = np.ascontiguousarray(eval_xyz, dtype=np.float64).reshape(-1, 3)
    nb = np.ascontiguousarray(nbr, dtype=np.int32)
    assert nb.ndim == 2 and nb.shape[0] == ev.shape[0]
    op = np.ascontiguousarray(ops, dtype=np.int32)
    out = np.zeros((len(op), ev.shape[0], nb.shape[1]))
    f = lib().mmg_rbf_weights
    f.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _dp, C.c_longlong, _dp, _ip, C.c_int, _ip, _dp]
    check(f(dim, polydeg, float(rbf_exp), nb.shape[1], cloud.shape[0], cloud.ctypes.data_as(_dp), ev.shape[0],
            ev.ctypes.data_as(_dp), nb.ctypes.data_as(_ip), len(op), op.ctypes.data_as(_ip), out.ctypes.data_as(_dp)))
    return out


def rbf_stencils(dim, polydeg, rbf_exp, stencil, cloud_xyz, eval_xyz, ops, cloud_flag=None, eval_flag=None, by_column=False):
    """mmg_rbf_stencils: neighbour search + weights in one call.  Returns (nbr [n_eval][stencil], weights
    [n_ops][n_eval][stencil], short_rows); rows nearest first, or in ascending neighbour id with by_column."""
    cloud = np.ascontiguousarray(cloud_xyz, dtype=np.float64).reshape(-1, 3)
    ev = np.ascontiguousarray(eval_xyz, dtype=np.float64).reshape(-1, 3)
    cf = None if cloud_flag is None else np.ascontiguousarray(cloud_flag, dtype=np.uint8)
    qf = None if eval_flag is None else np.ascontiguousarray(eval_flag, dtype=np.uint8)
    op = np.ascontiguousarray(ops, dtype=np.int32)
    nbr = np.zeros((ev.shape[0], int(stencil)), dtype=np.int32)
    w = np.zeros((len(op), ev.shape[0], int(stencil)))
    short = C.c_int(0)
    f = lib().mmg_rbf_stencils
    _bp = C.POINTER(C.c_ubyte)
    f.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _bp, C.c_longlong, _dp, _bp, C.c_int, _ip, C.c_int, _ip, _dp,
                  C.POINTER(C.c_int)]
    check(f(dim, polydeg, float(rbf_exp), int(stencil), cloud.shape[0], cloud.ctypes.data_as(_dp),
            cf.ctypes.data_as(_bp) if cf is not None else None, ev.shape[0], ev.ctypes.data_as(_dp),
            qf.ctypes.data_as(_bp) if qf is not None else None, len(op), op.ctypes.data_as(_ip), 1 if by_column else 0,
            nbr.ctypes.data_as(_ip), w.ctypes.data_as(_dp), C.byref(short)))
    return nbr, w, short.value


def knn(dim, cloud_xyz, query_xyz, k, cloud_flag=None, query_flag=None):
    """mmg_knn: indices of the k smallest (distance, index) pairs per query, [n_query][k] (-1 = cloud ran out)."""
    cloud = np.ascontiguousarray(cloud_xyz, dtype=np.float64).reshape(-1, 3)
    q = np.ascontiguousarray(query_xyz, dtype=np.float64).reshape(-1, 3)
    cf = None if cloud_flag is None else np.ascontiguousarray(cloud_flag, dtype=np.uint8)
    qf = None if query_flag is None else np.ascontiguousarray(query_flag, dtype=np.uint8)
    assert cf is None or len(cf) == len(cloud)
    assert qf is None or len(qf) == len(q)
    out = np.zeros((q.shape[0], int(k)), dtype=np.int32)
    f = lib().mmg_knn
    _bp = C.POINTER(C.c_ubyte)
    f.argtypes = [C.c_int, C.c_int, _dp, _bp, C.c_longlong, _dp, _bp, C.c_int, _ip]
    check(f(dim, cloud.shape[0], cloud.ctypes.data_as(_dp), cf.ctypes.data_as(_bp) if cf is not None else None, q.shape[0],
            q.ctypes.data_as(_dp), qf.ctypes.data_as(_bp) if qf is not None else None, int(k), out.ctypes.data_as(_ip)))
    return out


def device_count():
    n = C.c_int(0)
    lib().mmg_device_count(C.byref(n))
    return n.value


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return a.ctypes.data_as(_ip) if a is not None and a.size else None


def _pd(a):
    return a.ctypes.data_as(_dp) if a is not None and a.size else None


def make_desc(n, rowptr, col, val, bcflags, neumann, omega, iters, btype, bptr, bpts, bvals,
              tile_ptr=None, tile_size=0, lanes_per_row=0, tile_phase=None, waves_per_tile=0):
    """Returns (LevelDesc, keepalive) -- keepalive holds the numpy arrays."""
    keep = dict(rowptr=_i(rowptr), col=_i(col), val=_d(val), bcflags=_i(bcflags), btype=_i(btype),
                bptr=_i(bptr), bpts=_i(bpts), bvals=_d(bvals))
    d = LevelDesc()
    d.n = int(n)
    d.a_size = len(keep["rowptr"]) - 1
    d.rowptr, d.col, d.val = _pi(keep["rowptr"]), _pi(keep["col"]), _pd(keep["val"])
    d.bcflags = _pi(keep["bcflags"])
    d.neumann_flag = int(bool(neumann))
    d.omega, d.iters = float(omega), int(iters)
    d.nb = len(keep["btype"])
    d.btype, d.bptr = _pi(keep["btype"]), keep["bptr"].ctypes.data_as(_ip)
    d.bpts, d.bvals = _pi(keep["bpts"]), _pd(keep["bvals"])
    if tile_ptr is not None:
        keep["tile_ptr"] = _i(tile_ptr)
        d.tile_ptr = _pi(keep["tile_ptr"])
        d.n_tiles = len(keep["tile_ptr"]) - 1
    if tile_phase is not None and tile_ptr is not None:
        keep["tile_phase"] = _i(tile_phase)
        assert len(keep["tile_phase"]) == d.n_tiles
        d.tile_phase = _pi(keep["tile_phase"])
    d.tile_size = int(tile_size)
    d.lanes_per_row = int(lanes_per_row)
    d.waves_per_tile = int(waves_per_tile)
    return d, keep


class Level:
    """Device-side counterpart of one reference `Grid` (hot methods only)."""

    @classmethod
    def borrow(cls, handle, n, a_size):
        """Wrap an mmg_level* owned by someone else (a host C++ Grid)."""
        self = cls.__new__(cls)
        self.h = C.c_void_p(handle)
        self.n, self.a_size = int(n), int(a_size)
        self._borrowed = True
        return self

    def __init__(self, n, rowptr, col, val, bcflags, neumann, omega, iters, btype, bptr, bpts, bvals,
                 x=None, b=None, tile_ptr=None, tile_size=0, lanes_per_row=0, tile_phase=None, waves_per_tile=0):
        d, keep = make_desc(n, rowptr, col, val, bcflags, neumann, omega, iters, btype, bptr, bpts, bvals,
                            tile_ptr, tile_size, lanes_per_row, tile_phase, waves_per_tile)
        self.n, self.a_size = d.n, d.a_size
        self.h = C.c_void_p()
        check(lib().mmg_level_create(C.byref(self.h), C.byref(d)))
        if x is not None:
            self.set_x(x)
        if b is not None:
            self.set_rhs(b)

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None and not getattr(self, "_borrowed", False):
            _lib.mmg_level_destroy(self.h)
            self.h = None

    def set_exchange(self, n_owned, nbr_rank, send_ptr, send_idx, recv_ptr):
        nbr, sp, si, rp = _i(nbr_rank), _i(send_ptr), _i(send_idx), _i(recv_ptr)
        check(lib().mmg_level_set_exchange(self.h, int(n_owned), len(nbr), _pi(nbr), sp.ctypes.data_as(_ip),
                                           si.ctypes.data_as(_ip) if si.size else None, rp.ctypes.data_as(_ip)))

    def set_exchange_mode(self, per_phase):
        """collective; raises MmgError (mode unchanged) when the exact schedule does not exist for this partition"""
        check(lib().mmg_level_set_exchange_mode(self.h, int(per_phase)))

    def point_phases(self):
        ph = np.zeros(self.n, dtype=np.int32)
        check(lib().mmg_level_point_phases(self.h, ph.ctypes.data_as(_ip), self.n))
        return ph

    def exchange(self):
        check(lib().mmg_level_exchange(self.h))

    def exchange_info(self):
        """(neighbours, values sent, values received) of one ghost refresh"""
        k, ns, nr = C.c_int(0), C.c_longlong(0), C.c_longlong(0)
        check(lib().mmg_level_exchange_info(self.h, C.byref(k), C.byref(ns), C.byref(nr)))
        return k.value, ns.value, nr.value

    def time_exchange(self, reps):
        """device ms of `reps` ghost refreshes (collective)"""
        ms = (C.c_float * reps)()
        check(lib().mmg_level_time_exchange(self.h, int(reps), ms))
        return [float(v) for v in ms]

    def time_phases(self, nsweeps):
        ms, cnt = C.c_float(0), C.c_int(0)
        check(lib().mmg_level_time_phases(self.h, int(nsweeps), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def info(self):
        inf = LevelInfo()
        check(lib().mmg_level_info_get(self.h, C.byref(inf)))
        return inf.as_dict()

    def set_x(self, x):
        x = _d(x)
        check(lib().mmg_level_set_x(self.h, _pd(x), len(x)))

    def get_x(self):
        x = np.zeros(self.a_size)
        check(lib().mmg_level_get_x(self.h, _pd(x), len(x)))
        return x

    def set_rhs(self, b):
        b = _d(b)
        check(lib().mmg_level_set_rhs(self.h, _pd(b), len(b)))

    def get_rhs(self):
        b = np.zeros(self.a_size)
        check(lib().mmg_level_get_rhs(self.h, _pd(b), len(b)))
        return b

    def set_bvals(self, bvals):
        v = _d(bvals)
        check(lib().mmg_level_set_bvals(self.h, _pd(v), len(v)))

    def sor(self):
        check(lib().mmg_level_sor(self.h))

    def sweeps(self, k):
        check(lib().mmg_level_sweeps(self.h, int(k)))

    def bound_eval_neumann(self):
        check(lib().mmg_level_bound_eval_neumann(self.h))

    def residual(self):
        r = np.zeros(self.a_size)
        check(lib().mmg_level_residual(self.h, _pd(r), len(r)))
        return r

    def residual_ratio(self):
        v = C.c_double(0)
        check(lib().mmg_level_residual_ratio(self.h, C.byref(v)))
        return v.value

    def boundary_op(self, coarse):
        check(lib().mmg_level_boundary_op(self.h, int(coarse)))

    def modify_coeff_neumann(self, coarse):
        check(lib().mmg_level_modify_coeff_neumann(self.h, int(coarse)))

    def zero_x(self):
        check(lib().mmg_level_zero_x(self.h))

    def time_sweeps(self, nsweeps, reps):
        ms = np.zeros(reps, dtype=np.float32)
        check(lib().mmg_level_time_sweeps(self.h, int(nsweeps), int(reps), ms.ctypes.data_as(_fp)))
        return ms

    def time_residual(self, reps):
        ms = np.zeros(reps, dtype=np.float32)
        check(lib().mmg_level_time_residual(self.h, int(reps), ms.ctypes.data_as(_fp)))
        return ms


class Transfer:
    def __init__(self, rows, cols, outer, inner, val, col_major=True):
        outer, inner, val = _i(outer), _i(inner), _d(val)
        self.h = C.c_void_p()
        check(lib().mmg_transfer_create(C.byref(self.h), int(rows), int(cols), _pi(outer), _pi(inner), _pd(val),
                                        int(col_major)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mmg_transfer_destroy(self.h)
            self.h = None


def restrict(fine: Level, coarse: Level, R: Transfer):
    check(lib().mmg_restrict(fine.h, coarse.h, R.h))


def prolong_add(coarse: Level, fine: Level, P: Transfer):
    check(lib().mmg_prolong_add(coarse.h, fine.h, P.h))


class Hierarchy:
    """Device-side counterpart of the reference `Multigrid` (coarse -> fine)."""

    def __init__(self, levels, R, P, frac_step=False):
        self.levels, self.R, self.P = list(levels), list(R), list(P)
        nl = len(levels)
        vp = C.c_void_p
        la = (vp * nl)(*[l.h for l in levels])
        ra = (vp * nl)(*[(r.h if r is not None else None) for r in R])
        pa = (vp * nl)(*[(p.h if p is not None else None) for p in P])
        self.h = C.c_void_p()
        check(lib().mmg_hierarchy_create(C.byref(self.h), la, nl, ra, pa, int(frac_step)))
        self.residuals = []

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mmg_hierarchy_destroy(self.h)
            self.h = None

    def vcycle(self):
        v = C.c_double(0)
        check(lib().mmg_vcycle(self.h, C.byref(v)))
        if v.value >= 0:
            self.residuals.append(v.value)
        return v.value

    def vcycles(self, n):
        res = np.zeros(n)
        ms = C.c_float(0)
        check(lib().mmg_vcycles(self.h, int(n), _pd(res), C.byref(ms)))
        self.residuals.extend(float(r) for r in res if r >= 0)
        return res, ms.value

    def residual(self):
        v = C.c_double(0)
        check(lib().mmg_hierarchy_residual(self.h, C.byref(v)))
        return v.value


class Spmv:
    def __init__(self, rows, cols, rowptr, col, val):
        rowptr, col, val = _i(rowptr), _i(col), _d(val)
        self.rows, self.cols = int(rows), int(cols)
        self.h = C.c_void_p()
        check(lib().mmg_spmv_create(C.byref(self.h), self.rows, self.cols, _pi(rowptr), _pi(col), _pd(val)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mmg_spmv_destroy(self.h)
            self.h = None

    def apply(self, x):
        x = _d(x)
        y = np.zeros(self.rows)
        check(lib().mmg_spmv_apply(self.h, _pd(x), len(x), _pd(y), len(y)))
        return y


class FracStep:
    """Device-side FractionalStepGrid ops around a pressure Level (fractionalStepGrid.cpp:101-154)."""
    U, V, U_HAT, V_HAT = 0, 1, 2, 3

    def __init__(self, level, dx, dy, lap, nx, ny, bpts):
        self.level, self.n = level, level.n
        arrs = []
        for (rp, col, val) in (dx, dy, lap):
            arrs += [_i(rp), _i(col), _d(val)]
        nx, ny, bpts = _d(nx), _d(ny), _i(bpts)
        self.h = C.c_void_p()
        check(lib().mmg_fracstep_create(C.byref(self.h), level.h, self.n, _pi(arrs[0]), _pi(arrs[1]), _pd(arrs[2]),
                                        _pi(arrs[3]), _pi(arrs[4]), _pd(arrs[5]), _pi(arrs[6]), _pi(arrs[7]),
                                        _pd(arrs[8]), _pd(nx), _pd(ny), _pi(bpts), len(bpts)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mmg_fracstep_destroy(self.h)
            self.h = None

    def set(self, which, w):
        w = _d(w)
        check(lib().mmg_fracstep_set(self.h, which, _pd(w), len(w)))

    def get(self, which):
        w = np.zeros(self.n)
        check(lib().mmg_fracstep_get(self.h, which, _pd(w), len(w)))
        return w

    def calc_hat(self, dt, mu, rho):
        check(lib().mmg_fracstep_calc_hat(self.h, dt, mu, rho))

    def set_ppe_source(self, dt, rho):
        check(lib().mmg_fracstep_set_ppe_source(self.h, dt, rho))

    def correct(self, dt, rho):
        check(lib().mmg_fracstep_correct(self.h, dt, rho))

    def residual(self):
        v = C.c_double(0)
        check(lib().mmg_fracstep_residual(self.h, C.byref(v)))
        return v.value
