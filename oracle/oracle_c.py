"""
oracle_c.py -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).

ctypes binding of oracle/libmmg_oracle.so (mmg_oracle.c, the plain-C
restatement of the reference's V-cycle hot path) and of oracle/_ref's build of
the reference's own Eigen-free sources.  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _Level(C.Structure):
    _fields_ = [("n", C.c_int), ("a_size", C.c_int), ("rowptr", _ip), ("col", _ip), ("val", _dp),
                ("x", _dp), ("b", _dp), ("bcflags", _ip), ("neumann_flag", C.c_int),
                ("omega", C.c_double), ("iters", C.c_int), ("nb", C.c_int), ("btype", _ip),
                ("bptr", _ip), ("bpts", _ip), ("bvals", _dp)]


class _Csr(C.Structure):
    _fields_ = [("rows", C.c_int), ("rowptr", _ip), ("col", _ip), ("val", _dp)]


class _Csc(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("colptr", _ip), ("rowidx", _ip), ("val", _dp)]


def build(fast=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    import sys
    # make's chatter goes to stderr: bench.py must print exactly one JSON line on stdout
    subprocess.run(["make", "-s", "-C", _HERE, "all"], check=True, stdout=sys.stderr)
    if fast:
        subprocess.run(["make", "-s", "-C", _HERE, os.path.join(_HERE, "libmmg_oracle_fast.so")], check=True,
                       stdout=sys.stderr)


_libs = {}


def lib(fast=False):
    name = "libmmg_oracle_fast.so" if fast else "libmmg_oracle.so"
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build(fast)
        L = C.CDLL(path)
        L.orc_vcycle.restype = C.c_double
        L.orc_vcycle.argtypes = [C.POINTER(_Level), C.c_int, C.POINTER(_Csc), C.POINTER(_Csc), C.c_int]
        L.orc_vcycle_damped.restype = C.c_double
        L.orc_vcycle_damped.argtypes = [C.POINTER(_Level), C.c_int, C.POINTER(_Csc), C.POINTER(_Csc), C.c_int, C.c_double]
        L.orc_mg_residual.restype = C.c_double
        L.orc_mg_residual.argtypes = [C.POINTER(_Level), _dp]
        L.orc_l1.restype = C.c_double
        for f in ("orc_sor", "orc_bound_eval_neumann"):
            getattr(L, f).argtypes = [C.POINTER(_Level)]
            getattr(L, f).restype = None
        L.orc_sor_sweeps.argtypes = [C.POINTER(_Level), C.c_int]
        L.orc_sor_sweeps.restype = None
        L.orc_boundary_op.argtypes = [C.POINTER(_Level), C.c_int]
        L.orc_modify_coeff_neumann.argtypes = [C.POINTER(_Level), C.c_int]
        L.orc_residual.argtypes = [C.POINTER(_Level), _dp]
        L.orc_fix_vector_bound_coarse.argtypes = [C.POINTER(_Level), _dp]
        L.orc_csc_spmv.argtypes = [C.POINTER(_Csc), _dp, _dp]
        L.orc_sor_hybrid.argtypes = [C.POINTER(_Level), _ip, C.c_int, C.c_int]
        L.orc_vcycle_hybrid.restype = C.c_double
        L.orc_vcycle_hybrid.argtypes = [C.POINTER(_Level), C.c_int, C.POINTER(_Csc), C.POINTER(_Csc), C.POINTER(_ip), C.c_int]
        pc = C.POINTER(_Csr)
        L.orc_fs_calc_hat.argtypes = [C.c_int, pc, pc, pc, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp, _dp]
        L.orc_fs_set_ppe_source.argtypes = [C.c_int, pc, pc, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _ip, C.c_int,
                                            _dp, _dp, _dp]
        L.orc_fs_correct.argtypes = [C.c_int, pc, pc, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp]
        L.orc_fs_residual.argtypes = [C.c_int, _dp, _dp]
        L.orc_fs_residual.restype = C.c_double
        L.orc_fs_calc_hat3.argtypes = [C.c_int, pc, pc, pc, pc, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
                                       _dp, _dp, _dp]
        L.orc_fs_set_ppe_source3.argtypes = [C.c_int, pc, pc, pc, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
                                             _ip, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_fs_correct3.argtypes = [C.c_int, pc, pc, pc, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp]
        L.orc_push_inhomog.argtypes = [C.c_int, pc, _dp, _ip, _dp]
        L.orc_sor_sweeps_tiled.argtypes = [C.POINTER(_Level), C.c_int, _ip, C.c_int, _ip, C.c_int, C.c_int]
        _libs[name] = L
    return _libs[name]


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pi(a):
    return a.ctypes.data_as(_ip)


def _pd(a):
    return a.ctypes.data_as(_dp)


class Level:
    """One reference `Grid` as seen by the hot path. x and b are owned numpy
    arrays that the C oracle mutates in place."""

    def __init__(self, n, rowptr, col, val, x, b, bcflags, neumann, omega, iters, btype, bptr, bpts, bvals,
                 fast=False):
        self.n = int(n)
        self.rowptr, self.col, self.val = _i(rowptr), _i(col), _d(val)
        self.a_size = len(self.rowptr) - 1
        self.x, self.b = _d(x).copy(), _d(b).copy()
        self.bcflags = _i(bcflags)
        self.neumann = int(bool(neumann))
        self.omega, self.iters = float(omega), int(iters)
        self.btype, self.bptr, self.bpts, self.bvals = _i(btype), _i(bptr), _i(bpts), _d(bvals)
        self.fast = fast
        assert len(self.x) == self.a_size and len(self.b) == self.a_size

    @classmethod
    def from_grid(cls, g, fast=False):
        """g: oracle.setup_oracle.Grid after build_laplacian()."""
        rowptr, col, val = g.csr
        btype, bptr, bpts, bvals = g.boundary_arrays()
        return cls(g.n, rowptr, col, val, g.values, g.source, g.bcflags, g.neumann, g.props.omega,
                   g.props.iters, btype, bptr, bpts, bvals, fast=fast)

    def struct(self):
        s = _Level()
        s.n, s.a_size = self.n, self.a_size
        s.rowptr, s.col, s.val = _pi(self.rowptr), _pi(self.col), _pd(self.val)
        s.x, s.b, s.bcflags = _pd(self.x), _pd(self.b), _pi(self.bcflags)
        s.neumann_flag, s.omega, s.iters = self.neumann, self.omega, self.iters
        s.nb = len(self.btype)
        s.btype, s.bptr, s.bpts, s.bvals = _pi(self.btype), _pi(self.bptr), _pi(self.bpts), _pd(self.bvals)
        return s

    # --- the reference's Grid hot methods ---
    def sor(self):
        s = self.struct()
        lib(self.fast).orc_sor(C.byref(s))

    def sor_sweeps(self, k):
        s = self.struct()
        lib(self.fast).orc_sor_sweeps(C.byref(s), int(k))

    def sor_sweeps_tiled(self, k, tile_ptr, tile_phase, nthreads):
        """colour-parallel sweeps over multicolour tiles (baseline only); bitwise sor_sweeps on Dirichlet levels"""
        s = self.struct()
        tp, ph = _i(tile_ptr), _i(tile_phase)
        rc = lib(self.fast).orc_sor_sweeps_tiled(C.byref(s), int(k), _pi(tp), len(tp) - 1, _pi(ph), int(ph.max()) + 1, int(nthreads))
        if rc:
            raise ValueError("level not eligible for the colour-parallel sweep")

    def sor_hybrid(self, part, nparts, k):
        s = self.struct()
        part = _i(part)
        lib(self.fast).orc_sor_hybrid(C.byref(s), _pi(part), int(nparts), int(k))

    def bound_eval_neumann(self):
        s = self.struct()
        lib(self.fast).orc_bound_eval_neumann(C.byref(s))

    def boundary_op(self, coarse):
        s = self.struct()
        lib(self.fast).orc_boundary_op(C.byref(s), int(coarse))

    def modify_coeff_neumann(self, coarse):
        s = self.struct()
        lib(self.fast).orc_modify_coeff_neumann(C.byref(s), int(coarse))

    def residual(self):
        s = self.struct()
        r = np.zeros(self.a_size)
        lib(self.fast).orc_residual(C.byref(s), _pd(r))
        return r

    def residual_ratio(self):
        s = self.struct()
        w = np.zeros(self.a_size)
        return float(lib(self.fast).orc_mg_residual(C.byref(s), _pd(w)))


class Transfer:
    def __init__(self, rows, cols, colptr, rowidx, val):
        self.rows, self.cols = int(rows), int(cols)
        self.colptr, self.rowidx, self.val = _i(colptr), _i(rowidx), _d(val)

    @classmethod
    def from_dict(cls, d):
        return cls(d["rows"], d["cols"], d["colptr"], d["rowidx"], d["val"])

    def struct(self):
        s = _Csc()
        s.rows, s.cols = self.rows, self.cols
        s.colptr, s.rowidx, s.val = _pi(self.colptr), _pi(self.rowidx), _pd(self.val)
        return s

    def apply(self, x):
        x = _d(x)
        y = np.zeros(self.rows)
        s = self.struct()
        lib().orc_csc_spmv(C.byref(s), _pd(x), _pd(y))
        return y


class Multigrid:
    """Reference `Multigrid` (multigrid.h) over oracle Levels, coarse -> fine."""

    def __init__(self, levels, R, P, frac_step=False):
        self.levels = list(levels)
        self.R = [r if r is not None else Transfer(0, 0, [0], [], []) for r in R]
        self.P = [p if p is not None else Transfer(0, 0, [0], [], []) for p in P]
        self.frac_step = frac_step
        self.residuals = []

    def vcycle(self):
        nl = len(self.levels)
        lv = (_Level * nl)(*[l.struct() for l in self.levels])
        Rs = (_Csc * nl)(*[r.struct() for r in self.R])
        Ps = (_Csc * nl)(*[p.struct() for p in self.P])
        fast = self.levels[0].fast
        theta = float(getattr(self, "damping", 1.0))
        r = float(lib(fast).orc_vcycle(lv, nl, Rs, Ps, int(self.frac_step)) if theta == 1.0
                  else lib(fast).orc_vcycle_damped(lv, nl, Rs, Ps, int(self.frac_step), theta))
        if r >= 0:
            self.residuals.append(r)
        return r

    def residual(self):
        return self.levels[-1].residual_ratio()

    def vcycle_hybrid(self, parts, nparts):
        """V-cycle with the multi-GPU relaxation schedule; parts[l] = owner of every point of level l."""
        nl = len(self.levels)
        lv = (_Level * nl)(*[l.struct() for l in self.levels])
        Rs = (_Csc * nl)(*[r.struct() for r in self.R])
        Ps = (_Csc * nl)(*[p.struct() for p in self.P])
        keep = [_i(p) for p in parts]
        pp = (_ip * nl)(*[_pi(k) for k in keep])
        r = float(lib().orc_vcycle_hybrid(lv, nl, Rs, Ps, pp, int(nparts)))
        self.residuals.append(r)
        return r


# --------------------------------------------------------------------------
# oracle/_ref : the reference's own Eigen-free sources (prebuilt .so)
# --------------------------------------------------------------------------
def ref_lib():
    path = os.path.join(_HERE, "_ref", "libref_utils.so")
    if not os.path.exists(path):
        return None
    L = C.CDLL(path)
    L.ref_points_from_msh.argtypes = [C.c_char_p, _dp, C.c_int]
    L.ref_points_from_txt.argtypes = [C.c_char_p, _dp, C.c_int]
    L.ref_bound_pts_conn.argtypes = [C.c_char_p, _ip, C.c_int, _ip]
    L.ref_write_vector_txt.argtypes = [_dp, C.c_int, C.c_char_p]
    L.ref_distance.restype = C.c_double
    L.ref_distance.argtypes = [_dp, _dp]
    L.ref_shifting_scaling.argtypes = [_dp, C.c_int, _dp, _dp]
    L.ref_rcm.argtypes = [_ip, _ip, C.c_int, _ip]
    return L


class FracStep:
    """fractionalStepGrid.cpp:101-154 on plain arrays (oracle)."""

    def __init__(self, n, dx, dy, lap, nx, ny, bpts):
        self.n = int(n)
        self._m = []
        for (rp, col, val) in (dx, dy, lap):
            rp, col, val = _i(rp), _i(col), _d(val)
            st = _Csr()
            st.rows, st.rowptr, st.col, st.val = self.n, _pi(rp), _pi(col), _pd(val)
            self._m.append((st, rp, col, val))
        self.nx, self.ny, self.bpts = _d(nx), _d(ny), _i(bpts)
        self.u, self.v = np.zeros(self.n), np.zeros(self.n)
        self.u_hat, self.v_hat = np.zeros(self.n), np.zeros(self.n)

    def calc_hat(self, dt, mu, rho):
        lib().orc_fs_calc_hat(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), C.byref(self._m[2][0]),
                              _pd(self.u), _pd(self.v), dt, mu, rho, _pd(self.u_hat), _pd(self.v_hat))

    def set_ppe_source(self, source, dt, rho):
        lib().orc_fs_set_ppe_source(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), _pd(self.u), _pd(self.v),
                                    _pd(self.u_hat), _pd(self.v_hat), dt, rho, _pi(self.bpts), len(self.bpts),
                                    _pd(self.nx), _pd(self.ny), _pd(source))

    def correct(self, p, dt, rho):
        p = _d(p)
        lib().orc_fs_correct(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), _pd(p), _pd(self.u_hat),
                             _pd(self.v_hat), dt, rho, _pd(self.u), _pd(self.v))

    def residual(self):
        return float(lib().orc_fs_residual(self.n, _pd(self.u), _pd(self.u_hat)))


class FracStep3(FracStep):
    """The 3-D extension (mmg_oracle.c: orc_fs_*3): third velocity component, D_z, n_z."""

    def __init__(self, n, dx, dy, dz, lap, nx, ny, nz, bpts):
        super().__init__(n, dx, dy, lap, nx, ny, bpts)
        rp, col, val = _i(dz[0]), _i(dz[1]), _d(dz[2])
        st = _Csr()
        st.rows, st.rowptr, st.col, st.val = self.n, _pi(rp), _pi(col), _pd(val)
        self._dz = (st, rp, col, val)
        self.nz = _d(nz)
        self.w, self.w_hat = np.zeros(self.n), np.zeros(self.n)

    def calc_hat(self, dt, mu, rho):
        lib().orc_fs_calc_hat3(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), C.byref(self._dz[0]),
                               C.byref(self._m[2][0]), _pd(self.u), _pd(self.v), _pd(self.w), dt, mu, rho,
                               _pd(self.u_hat), _pd(self.v_hat), _pd(self.w_hat))

    def set_ppe_source(self, source, dt, rho):
        lib().orc_fs_set_ppe_source3(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), C.byref(self._dz[0]),
                                     _pd(self.u), _pd(self.v), _pd(self.w), _pd(self.u_hat), _pd(self.v_hat),
                                     _pd(self.w_hat), dt, rho, _pi(self.bpts), len(self.bpts), _pd(self.nx), _pd(self.ny),
                                     _pd(self.nz), _pd(source))

    def correct(self, p, dt, rho):
        p = _d(p)
        lib().orc_fs_correct3(self.n, C.byref(self._m[0][0]), C.byref(self._m[1][0]), C.byref(self._dz[0]), _pd(p),
                              _pd(self.u_hat), _pd(self.v_hat), _pd(self.w_hat), dt, rho, _pd(self.u), _pd(self.v),
                              _pd(self.w))


def push_inhomog(n, bc_csr, diags, bcflags, source):
    """Grid::push_inhomog_to_rhs (grid.cpp:664-685) in place on `source` (float64 array of >= n entries)."""
    rp, col, val = _i(bc_csr[0]), _i(bc_csr[1]), _d(bc_csr[2])
    st = _Csr()
    st.rows, st.rowptr, st.col, st.val = int(n), _pi(rp), _pi(col), _pd(val)
    diags, bcflags = _d(diags), _i(bcflags)
    assert source.dtype == np.float64 and source.flags["C_CONTIGUOUS"]
    lib().orc_push_inhomog(int(n), C.byref(st), _pd(diags), _pi(bcflags), _pd(source))

