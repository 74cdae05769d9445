"""ctypes binding of libmmgp_host.so: the host C++ mirror of the reference's
Grid / Multigrid classes (csrc/host) through the C harness in csrc/host/harness.cpp.
Plumbing for tests and bench.py; no compute happens in Python."""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

_PKG = os.environ.get("MMGP_LIBDIR") or os.path.dirname(os.path.abspath(__file__))  # MMGP_LIBDIR: A/B builds
LIB_PATH = os.path.join(_PKG, "libmmgp_host.so")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)
_lib = None

ORDER_RCM, ORDER_MC, ORDER_NONE = 0, 1, 2
KIND_DIRICHLET, KIND_NEUMANN, KIND_GRAPH = 0, 1, 2


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostError(f"{LIB_PATH} not built (run __graft_entry__.build())")
        from . import _capi
        _capi.lib()  # libmmgp.so first (rpath also finds it)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.mmgh_last_error.restype = C.c_char_p
        L.mmgh_mg_create_square.restype = vp
        L.mmgh_mg_create_square.argtypes = [C.c_int, _ip, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_double, C.c_int, C.c_int, _dp, C.c_int]
        L.mmgh_mg_destroy.argtypes = [vp]
        L.mmgh_mg_nlevels.argtypes = [vp]
        L.mmgh_mg_grid.restype = vp
        L.mmgh_mg_grid.argtypes = [vp, C.c_int]
        L.mmgh_mg_vcycle.argtypes = [vp, _dp]
        L.mmgh_mg_vcycles.argtypes = [vp, C.c_int, _dp, _fp]
        L.mmgh_mg_residual.argtypes = [vp, _dp]
        L.mmgh_mg_transfer_shape.argtypes = [vp, C.c_int, C.c_int, _ip, _ip, _ip]
        L.mmgh_mg_transfer_get.argtypes = [vp, C.c_int, C.c_int, _ip, _ip, _dp]
        L.mmgh_grid_create_square.restype = vp
        L.mmgh_grid_create_square.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_double, C.c_int, C.c_int, C.c_int]
        L.mmgh_grid_destroy.argtypes = [vp]
        L.mmgh_grid_sizes.argtypes = [vp, _ip]
        L.mmgh_grid_get_csr.argtypes = [vp, _ip, _ip, _dp]
        L.mmgh_grid_get_points.argtypes = [vp, _dp, _ip]
        L.mmgh_grid_get_boundaries.argtypes = [vp, _ip, _ip, _ip, _dp]
        L.mmgh_grid_get_tile_ptr.argtypes = [vp, _ip]
        L.mmgh_grid_get_tile_phase.argtypes = [vp, _ip]
        L.mmgh_mg_setup_exchange.argtypes = [vp, C.c_int]
        for f in ("mmgh_grid_get_values", "mmgh_grid_get_source", "mmgh_grid_set_values", "mmgh_grid_set_source",
                  "mmgh_grid_residual", "mmgh_grid_residual_ratio"):
            getattr(L, f).argtypes = [vp, _dp]
        L.mmgh_grid_set_value_at.argtypes = [vp, C.c_int, C.c_double]
        L.mmgh_grid_value_at.argtypes = [vp, C.c_int]
        L.mmgh_grid_value_at.restype = C.c_double
        for f in ("mmgh_grid_sor", "mmgh_grid_bound_eval_neumann", "mmgh_grid_sor_wrong_args"):
            getattr(L, f).argtypes = [vp]
        for f in ("mmgh_grid_boundary_op", "mmgh_grid_modify_coeff_neumann"):
            getattr(L, f).argtypes = [vp, C.c_int]
        L.mmgh_grid_device.restype = vp
        L.mmgh_grid_device.argtypes = [vp]
        L.mmgh_distance.restype = C.c_double
        L.mmgh_distance.argtypes = [_dp, _dp]
        L.mmgh_shifting_scaling.argtypes = [_dp, C.c_int, _dp, _dp]
        L.mmgh_rcm.argtypes = [_ip, _ip, C.c_int, _ip]
        L.mmgh_points_from_msh.argtypes = [C.c_char_p, _dp, C.c_int, C.c_int]
        L.mmgh_bound_pts_conn.argtypes = [C.c_char_p, _ip, C.c_int, _ip]
        L.mmgh_write_vector_txt.argtypes = [_dp, C.c_int, C.c_char_p]
        L.mmgh_order_from_txt.argtypes = [C.c_char_p, C.c_int]
        L.mmgh_write_msh.argtypes = [C.c_char_p, _dp, C.c_int]
        L.mmgh_write_bin.argtypes = [C.c_char_p, _dp, C.c_int, C.c_int]
        L.mmgh_points_from_bin.argtypes = [C.c_char_p, _dp, C.c_longlong, _ip]
        L.mmgh_points_from_bin.restype = C.c_longlong
        L.mmgh_grid_knn.argtypes = [vp, C.c_int, C.c_int, _ip]
        L.mmgh_fs_create_square.restype = vp
        L.mmgh_fs_create_square.argtypes = [C.c_int, _dp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        if hasattr(L, "mmgh_mg_extract_subdomain_replicated"):
            L.mmgh_mg_extract_subdomain_replicated.restype = vp
            L.mmgh_mg_extract_subdomain_replicated.argtypes = [vp, C.c_int, C.c_int, C.c_int]
            L.mmgh_mg_gather_info.argtypes = [vp, _ip, _ip]
            L.mmgh_grid_is_replicated.argtypes = [vp]
        if hasattr(L, "mmgh_fs_create_box"):  # (absent from older builds loaded through MMGP_LIBDIR for A/B runs)
            L.mmgh_fs_create_box.restype = vp
            L.mmgh_fs_create_box.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
            L.mmgh_mg_create_fs.restype = vp
            L.mmgh_mg_create_fs.argtypes = [C.c_int, _ip, _dp, _ip, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
            L.mmgh_mg_fs_step.argtypes = [vp, C.c_int, _ip, _dp]
            L.mmgh_grid_coupling_nnz.argtypes = [vp]
            L.mmgh_grid_coupling_get.argtypes = [vp, _ip, _ip, _dp, _dp]
            L.mmgh_fs_get_normal_z.argtypes = [vp, _dp]
        L.mmgh_fs_op_nnz.argtypes = [vp, C.c_int]
        L.mmgh_fs_op_get.argtypes = [vp, C.c_int, _ip, _ip, _dp]
        L.mmgh_fs_get_normals.argtypes = [vp, _dp, _dp]
        L.mmgh_fs_get_vec.argtypes = [vp, C.c_int, _dp]
        L.mmgh_fs_set_vec.argtypes = [vp, C.c_int, _dp]
        for f in ("mmgh_fs_prescribe_soln", "mmgh_fs_set_uv_bound", "mmgh_fs_calc_hat", "mmgh_fs_set_ppe_source",
                  "mmgh_fs_push_inhomog", "mmgh_fs_correct"):
            getattr(L, f).argtypes = [vp]
        L.mmgh_fs_residual.argtypes = [vp, _dp]
        L.mmgh_mg_extract_subdomain.restype = vp
        L.mmgh_mg_extract_subdomain.argtypes = [vp, C.c_int, C.c_int]
        L.mmgh_mg_level_part.argtypes = [vp, C.c_int, C.c_int, _ip]
        L.mmgh_grid_partition_slabs.argtypes = [vp, C.c_int, _ip]
        if hasattr(L, "mmgh_grid_partition"):
            L.mmgh_grid_partition.argtypes = [vp, C.c_int, _ip]
        L.mmgh_grid_extract_subdomain.restype = vp
        L.mmgh_grid_extract_subdomain.argtypes = [vp, _ip, C.c_int]
        L.mmgh_grid_n_owned.argtypes = [vp]
        L.mmgh_grid_local_map.argtypes = [vp, _ip, _ip]
        L.mmgh_grid_create_local.restype = vp
        L.mmgh_grid_create_local.argtypes = [C.c_int, _dp, _ip, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_int, C.c_int, C.c_int]
        L.mmgh_set_option.argtypes = [C.c_char_p, C.c_int]
        _lib = L
    return _lib


def set_option(name, value):
    """mmgh_set_option: "device_setup" -1 automatic / 0 host threads / 1 batched dense solves on the MI355X."""
    if lib().mmgh_set_option(name.encode(), int(value)) != 0:
        raise HostError(_err())


def write_cloud_bin(fname, points, dim):
    """Binary point-cloud container (fileReadingFunctions.h: writePointsToBinFile)."""
    pts = _d(points).reshape(-1, 3)
    if lib().mmgh_write_bin(os.fsencode(fname), pts.ctypes.data_as(_dp), len(pts), int(dim)) != 0:
        raise HostError(f"cannot write {fname}")


def read_cloud_bin(fname):
    """-> (points[n, 3], dim); raises on a missing / foreign / truncated file."""
    d = C.c_int(0)
    n = lib().mmgh_points_from_bin(os.fsencode(fname), None, 0, C.byref(d))
    if n <= 0:
        raise HostError(f"{fname}: not a readable MMGCLOUD file")
    xyz = np.zeros((n, 3))
    lib().mmgh_points_from_bin(os.fsencode(fname), xyz.ctypes.data_as(_dp), n, C.byref(d))
    return xyz, d.value


def stencil_size(polydeg, dim=2):
    """int(2.5 * polyTerms) -- grid.cpp:266-267 (3-D: (L+1)(L+2)(L+3)/6 terms)."""
    pt = (polydeg + 1) * (polydeg + 2) * (polydeg + 3) // 6 if dim >= 3 else (polydeg + 1) * (polydeg + 2) // 2
    return int(2.5 * pt)


def _err():
    return lib().mmgh_last_error().decode()


def _chk(rc):
    if rc != 0:
        raise HostError(_err())


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Grid:
    """Handle on a C++ `Grid` (csrc/host/grid.h)."""

    def __init__(self, handle, owner=None):
        self.h = handle
        self._owner = owner  # a Multigrid owns its grids (multigrid.cpp:10-16)

    @classmethod
    def create_square(cls, points, polydeg, dim=2, kind=KIND_DIRICHLET, k1=1, k2=1, ordering=ORDER_MC,
                      tile_points=512, omega=1.4, iters=5, lanes_per_row=0, stencil=0):
        pts = _d(points).reshape(-1, 3)
        h = lib().mmgh_grid_create_square(len(pts), pts.ctypes.data_as(_dp), polydeg, dim, kind, k1, k2, ordering,
                                          tile_points, omega, iters, lanes_per_row, stencil)
        if not h:
            raise HostError(_err())
        g = cls(h)
        g._own = True
        return g

    def __del__(self):
        if getattr(self, "_own", False) and self.h and _lib is not None:
            _lib.mmgh_grid_destroy(self.h)
            self.h = None

    # ---- domain decomposition ---------------------------------------------------------
    def partition_slabs(self, nparts):
        part = np.zeros(self.sizes()["n"], dtype=np.int32)
        lib().mmgh_grid_partition_slabs(self.h, int(nparts), part.ctypes.data_as(_ip))
        return part

    def extract_subdomain(self, part, rank):
        part = _i(part)
        h = lib().mmgh_grid_extract_subdomain(self.h, part.ctypes.data_as(_ip), int(rank))
        if not h:
            raise HostError(_err())
        g = Grid(h)
        g._own = True
        return g

    @classmethod
    def create_local(cls, points, flags, gid, owner, dim, stencil, tile_points=0, lanes_per_row=0, omega=1.4, iters=5,
                     kind=KIND_GRAPH, polydeg=3):
        pts, flags, gid, owner = _d(points).reshape(-1, 3), _i(flags), _i(gid), _i(owner)
        h = lib().mmgh_grid_create_local(len(pts), pts.ctypes.data_as(_dp), flags.ctypes.data_as(_ip),
                                         gid.ctypes.data_as(_ip), owner.ctypes.data_as(_ip), dim, stencil, tile_points,
                                         lanes_per_row, omega, iters, kind, polydeg)
        if not h:
            raise HostError(_err())
        g = cls(h)
        g._own = True
        return g

    def is_replicated(self):
        return bool(lib().mmgh_grid_is_replicated(self.h))

    def local_map(self):
        """(n_owned, gid[n_local], ghost_owner[n_ghost]) of a sub-domain grid."""
        n = self.sizes()["n"]
        no = lib().mmgh_grid_n_owned(self.h)
        gid = np.zeros(n, dtype=np.int32)
        own = np.zeros(max(n - no, 1), dtype=np.int32)
        lib().mmgh_grid_local_map(self.h, gid.ctypes.data_as(_ip), own.ctypes.data_as(_ip))
        return no, gid, own[: n - no]

    def sizes(self):
        out = np.zeros(8, dtype=np.int32)
        lib().mmgh_grid_sizes(self.h, out.ctypes.data_as(_ip))
        return dict(zip(["n", "a_size", "nnz", "neumann", "nb", "nbpts", "n_tiles", "stencil"], out.tolist()))

    def relaxation(self):
        """(omega, iters) of this grid's GridProperties."""
        w, it = C.c_double(0), C.c_int(0)
        f = lib().mmgh_grid_get_relaxation
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
        f(self.h, C.byref(w), C.byref(it))
        return w.value, it.value

    def set_relaxation(self, omega, iters):
        """GridProperties::omega / iters of this grid alone (e.g. a coarse grid relaxed longer than the fine ones)."""
        f = lib().mmgh_grid_set_relaxation
        f.argtypes = [C.c_void_p, C.c_double, C.c_int]
        f(self.h, float(omega), int(iters))

    def csr(self):
        s = self.sizes()
        rowptr = np.zeros(s["a_size"] + 1, dtype=np.int32)
        col = np.zeros(s["nnz"], dtype=np.int32)
        val = np.zeros(s["nnz"])
        lib().mmgh_grid_get_csr(self.h, rowptr.ctypes.data_as(_ip), col.ctypes.data_as(_ip), val.ctypes.data_as(_dp))
        return rowptr, col, val

    def points(self):
        s = self.sizes()
        xyz = np.zeros((s["n"], 3))
        f = np.zeros(s["n"], dtype=np.int32)
        lib().mmgh_grid_get_points(self.h, xyz.ctypes.data_as(_dp), f.ctypes.data_as(_ip))
        return xyz, f

    def boundaries(self):
        s = self.sizes()
        btype = np.zeros(s["nb"], dtype=np.int32)
        bptr = np.zeros(s["nb"] + 1, dtype=np.int32)
        bpts = np.zeros(s["nbpts"], dtype=np.int32)
        bvals = np.zeros(s["nbpts"])
        lib().mmgh_grid_get_boundaries(self.h, btype.ctypes.data_as(_ip), bptr.ctypes.data_as(_ip),
                                       bpts.ctypes.data_as(_ip), bvals.ctypes.data_as(_dp))
        return btype, bptr, bpts, bvals

    def tile_ptr(self):
        s = self.sizes()
        if s["n_tiles"] == 0:
            return None
        tp = np.zeros(s["n_tiles"] + 1, dtype=np.int32)
        lib().mmgh_grid_get_tile_ptr(self.h, tp.ctypes.data_as(_ip))
        return tp

    def exchange_lists(self):
        """(nbr, send_ptr, send_idx, recv_ptr) worked out by Multigrid::extract_subdomain in C++, or None."""
        L = lib()
        L.mmgh_grid_exchange_lists.argtypes = [C.c_void_p, _ip, _ip, _ip, _ip, _ip]
        sz = np.zeros(2, dtype=np.int32)
        if L.mmgh_grid_exchange_lists(self.h, sz.ctypes.data_as(_ip), None, None, None, None) != 0:
            return None
        nbr = np.zeros(sz[0], dtype=np.int32)
        sp = np.zeros(sz[0] + 1, dtype=np.int32)
        si = np.zeros(max(sz[1], 1), dtype=np.int32)
        rp = np.zeros(sz[0] + 1, dtype=np.int32)
        L.mmgh_grid_exchange_lists(self.h, sz.ctypes.data_as(_ip), nbr.ctypes.data_as(_ip), sp.ctypes.data_as(_ip),
                                   si.ctypes.data_as(_ip), rp.ctypes.data_as(_ip))
        return nbr, sp, si[:sz[1]], rp

    def tile_phase(self):
        """Phase numbers (global tile colours) a sub-domain grid passes to libmmgp, or None."""
        n = lib().mmgh_grid_get_tile_phase(self.h, None)
        if n == 0:
            return None
        tc = np.zeros(n, dtype=np.int32)
        lib().mmgh_grid_get_tile_phase(self.h, tc.ctypes.data_as(_ip))
        return tc

    def values(self):
        x = np.zeros(self.sizes()["a_size"])
        _chk(lib().mmgh_grid_get_values(self.h, x.ctypes.data_as(_dp)))
        return x

    def source(self):
        b = np.zeros(self.sizes()["a_size"])
        _chk(lib().mmgh_grid_get_source(self.h, b.ctypes.data_as(_dp)))
        return b

    def set_values(self, x):
        x = _d(x)
        assert len(x) == self.sizes()["a_size"]
        lib().mmgh_grid_set_values(self.h, x.ctypes.data_as(_dp))

    def set_source(self, b):
        b = _d(b)
        assert len(b) == self.sizes()["a_size"]
        lib().mmgh_grid_set_source(self.h, b.ctypes.data_as(_dp))

    def level_arrays(self, omega=1.4, iters=5):
        """Everything the oracle / the raw C-ABI need, as numpy arrays."""
        s = self.sizes()
        rowptr, col, val = self.csr()
        _xyz, flags = self.points()
        btype, bptr, bpts, bvals = self.boundaries()
        return dict(n=s["n"], a_size=s["a_size"], rowptr=rowptr, col=col, val=val, bcflags=flags,
                    neumann=s["neumann"], omega=omega, iters=iters, btype=btype, bptr=bptr, bpts=bpts, bvals=bvals,
                    x0=self.values(), b0=self.source())

    # hot methods (device)
    def sor(self):
        _chk(lib().mmgh_grid_sor(self.h))

    def boundary_op(self, coarse):
        _chk(lib().mmgh_grid_boundary_op(self.h, int(coarse)))

    def bound_eval_neumann(self):
        _chk(lib().mmgh_grid_bound_eval_neumann(self.h))

    def modify_coeff_neumann(self, coarse):
        _chk(lib().mmgh_grid_modify_coeff_neumann(self.h, int(coarse)))

    def residual(self):
        r = np.zeros(self.sizes()["a_size"])
        _chk(lib().mmgh_grid_residual(self.h, r.ctypes.data_as(_dp)))
        return r

    def residual_ratio(self):
        v = C.c_double(0)
        _chk(lib().mmgh_grid_residual_ratio(self.h, C.byref(v)))
        return v.value

    def device_level(self):
        """Raw mmg_level* (void*) for direct C-ABI calls (bench timing)."""
        d = lib().mmgh_grid_device(self.h)
        if not d:
            raise HostError(_err())
        return d

    def knn(self, pid, k):
        out = np.zeros(k, dtype=np.int32)
        n = lib().mmgh_grid_knn(self.h, pid, k, out.ctypes.data_as(_ip))
        return out[:n]


class FracStepGrid(Grid):
    """Handle on a C++ `FractionalStepGrid` (csrc/host/fractionalStepGrid.hpp), built by the
    call sequence of genFractionalStepGrid (FractionalStepSim.cpp:3-49)."""

    @classmethod
    def create(cls, points, polydeg=3, dt=2e-4, mu=0.025, rho=1.0, ordering=ORDER_MC, tile_points=0, coarse=False, dim=2):
        """dim 2: the reference's Kovasznay set-up (FractionalStepSim.cpp:3-49); dim 3: the 3-D extension on a
        box cloud (third velocity component, D_z; Taylor-Green-shaped velocity data)."""
        pts = _d(points).reshape(-1, 3)
        if dim == 2:
            h = lib().mmgh_fs_create_square(len(pts), pts.ctypes.data_as(_dp), polydeg, dt, mu, rho, ordering, tile_points,
                                            int(coarse))
        else:
            h = lib().mmgh_fs_create_box(len(pts), pts.ctypes.data_as(_dp), int(dim), polydeg, dt, mu, rho, ordering,
                                         tile_points, int(coarse))
        if not h:
            raise HostError(_err())
        g = cls(h)
        g._own = True
        g.dt, g.mu, g.rho, g.dim = dt, mu, rho, dim
        return g

    def normal_z(self):
        nz = np.zeros(self.sizes()["n"])
        lib().mmgh_fs_get_normal_z(self.h, nz.ctypes.data_as(_dp))
        return nz

    def coupling(self):
        """(rowptr, col, val), diag: neumann_boundary_coeffs_ and diags behind Grid::push_inhomog_to_rhs."""
        n = self.sizes()["n"]
        nnz = lib().mmgh_grid_coupling_nnz(self.h)
        rp, col, val, dg = np.zeros(n + 2, dtype=np.int32), np.zeros(max(nnz, 1), dtype=np.int32), np.zeros(max(nnz, 1)), np.zeros(n)
        lib().mmgh_grid_coupling_get(self.h, rp.ctypes.data_as(_ip), col.ctypes.data_as(_ip), val.ctypes.data_as(_dp),
                                     dg.ctypes.data_as(_dp))
        return (rp[:n + 1], col[:nnz], val[:nnz]), dg

    def op(self, which):
        n = self.sizes()["n"]
        nnz = lib().mmgh_fs_op_nnz(self.h, which)
        rp, col, val = np.zeros(n + 1, dtype=np.int32), np.zeros(nnz, dtype=np.int32), np.zeros(nnz)
        lib().mmgh_fs_op_get(self.h, which, rp.ctypes.data_as(_ip), col.ctypes.data_as(_ip), val.ctypes.data_as(_dp))
        return rp, col, val

    def normals(self):
        n = self.sizes()["n"]
        nx, ny = np.zeros(n), np.zeros(n)
        lib().mmgh_fs_get_normals(self.h, nx.ctypes.data_as(_dp), ny.ctypes.data_as(_dp))
        return nx, ny

    def vec(self, which):
        w = np.zeros(self.sizes()["n"])
        _chk(lib().mmgh_fs_get_vec(self.h, which, w.ctypes.data_as(_dp)))
        return w

    def set_vec(self, which, w):
        w = _d(w)
        lib().mmgh_fs_set_vec(self.h, which, w.ctypes.data_as(_dp))

    def prescribe_soln(self):
        lib().mmgh_fs_prescribe_soln(self.h)

    def set_uv_bound(self):
        lib().mmgh_fs_set_uv_bound(self.h)

    def calc_hat(self):
        _chk(lib().mmgh_fs_calc_hat(self.h))

    def set_ppe_source(self):
        _chk(lib().mmgh_fs_set_ppe_source(self.h))

    def push_inhomog_to_rhs(self):
        _chk(lib().mmgh_fs_push_inhomog(self.h))

    def correct(self):
        _chk(lib().mmgh_fs_correct(self.h))

    def fs_residual(self):
        v = C.c_double(0)
        _chk(lib().mmgh_fs_residual(self.h, C.byref(v)))
        return v.value


class Multigrid:
    """Handle on a C++ `Multigrid` built by the reference's factory sequence."""

    def __init__(self, clouds, polydegs, dim=2, neumann=False, k1=1, k2=1, ordering=ORDER_MC, tile_points=512,
                 omega=1.4, iters=5, frac_step=False, bval_abc=None, lanes_per_row=0):
        npts = _i([len(c) for c in clouds])
        xyz = _d(np.concatenate([_d(c).reshape(-1, 3) for c in clouds], axis=0))
        pd = _i(polydegs)
        abc = _d(bval_abc) if bval_abc is not None else None
        self.h = lib().mmgh_mg_create_square(len(clouds), npts.ctypes.data_as(_ip), xyz.ctypes.data_as(_dp),
                                             pd.ctypes.data_as(_ip), dim, int(neumann), k1, k2, ordering, tile_points,
                                             omega, iters, int(frac_step),
                                             abc.ctypes.data_as(_dp) if abc is not None else None, lanes_per_row)
        if not self.h:
            raise HostError(_err())
        self.omega, self.iters, self.frac_step = omega, iters, bool(frac_step)
        self.residuals = []

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.mmgh_mg_destroy(self.h)
            self.h = None

    @classmethod
    def square_with_circle(cls, clouds, polydegs, k=1, ordering=ORDER_MC, tile_points=0, omega=1.4, iters=5):
        """The reference's "square_with_circle" test problem (testing_functions.cpp:85-106): unit square with a hole of
        radius 0.25, u = 0 on the square, u = sin(k pi x) sin(k pi y) on the circle (inhomogeneous second boundary)."""
        return cls._geom(1, clouds, polydegs, k, ordering, tile_points, omega, iters)

    @classmethod
    def annulus(cls, clouds, polydegs, k=1, ordering=ORDER_MC, tile_points=0, omega=1.4, iters=5):
        """The reference's "concentric_circles" test problem (testing_functions.cpp:107-135): annulus 0.25 <= r <= 0.5
        around (0.5, 0.5), homogeneous Dirichlet data on both circles, manufactured solution sin(pi k r*)."""
        return cls._geom(2, clouds, polydegs, k, ordering, tile_points, omega, iters)

    @classmethod
    def annulus_neumann(cls, clouds, polydegs, k=1, ordering=ORDER_MC, tile_points=0, omega=1.4, iters=5):
        """"concentric_circles" with Neumann data on both circles (testing_functions.cpp:212-250): the normal derivative
        of sin(pi k r*) along the inward normals, non-zero on both boundaries (push_inhomog_to_rhs)."""
        return cls._geom(3, clouds, polydegs, k, ordering, tile_points, omega, iters)

    @classmethod
    def square_with_circle_neumann(cls, clouds, polydegs, k=1, ordering=ORDER_MC, tile_points=0, omega=1.4, iters=5):
        """"square_with_circle" with Neumann data (testing_functions.cpp:186-209): cos cos, zero normal derivative on
        the square, its derivative along the radial normal on the circle."""
        return cls._geom(4, clouds, polydegs, k, ordering, tile_points, omega, iters)

    @classmethod
    def _geom(cls, geom, clouds, polydegs, k, ordering, tile_points, omega, iters):
        npts = _i([len(c) for c in clouds])
        xyz = _d(np.concatenate([_d(c).reshape(-1, 3) for c in clouds], axis=0))
        pd = _i(polydegs)
        f = lib().mmgh_mg_create_geom
        f.restype = C.c_void_p
        f.argtypes = [C.c_int, C.c_int, _ip, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]
        h = f(int(geom), len(clouds), npts.ctypes.data_as(_ip), xyz.ctypes.data_as(_dp), pd.ctypes.data_as(_ip), int(k), ordering,
              tile_points, float(omega), int(iters))
        if not h:
            raise HostError(_err())
        return cls._from_handle(h, omega, iters)

    def set_correction_damping(self, theta):
        """NOT in the reference (opt-in safeguard): x_f += theta * P x_c instead of the plain coarse-grid correction."""
        f = lib().mmgh_mg_set_correction_damping
        f.argtypes = [C.c_void_p, C.c_double]
        _chk(f(self.h, float(theta)))
        self.damping = float(theta)

    @staticmethod
    def last_setup_times():
        """Wall seconds of the stages of the last constructor call: one entry per grid (cloud -> ordering ->
        operator), then Multigrid::buildMatrices."""
        f = lib().mmgh_mg_setup_times
        f.restype = C.c_int
        f.argtypes = [_dp, C.c_int]
        n = f(None, 0)
        out = np.zeros(max(n, 1))
        f(out.ctypes.data_as(_dp), n)
        return [float(v) for v in out[:n]]

    @classmethod
    def _from_handle(cls, h, omega, iters):
        self = cls.__new__(cls)
        self.h, self.omega, self.iters, self.residuals, self.frac_step = h, omega, iters, [], False
        return self

    def extract_subdomain(self, nparts, rank, replicate_below=0):
        """Rank's sub-domain of every level + the local rows of the transfers (host C++).  replicate_below > 0:
        coarse levels of at most that many points stay complete on every rank (no exchange on them; the
        restriction into them reads the all-gathered residual of the coarsest decomposed level)."""
        if replicate_below > 0:
            h = lib().mmgh_mg_extract_subdomain_replicated(self.h, int(nparts), int(rank), int(replicate_below))
        else:
            h = lib().mmgh_mg_extract_subdomain(self.h, int(nparts), int(rank))
        if not h:
            raise HostError(_err())
        sub = type(self)._from_handle(h, self.omega, self.iters)
        sub.frac_step = bool(getattr(self, "frac_step", False))
        for k in ("dt", "mu", "rho", "dim", "damping"):      # FracStepMultigrid: the flow parameters travel along
            if hasattr(self, k):
                setattr(sub, k, getattr(self, k))
        return sub

    def gather_info(self):
        """(level, ranks, max_count, n_global, gid[ranks, max_count]) of a hierarchy with replicated coarse levels, or None."""
        o = np.zeros(4, dtype=np.int32)
        lib().mmgh_mg_gather_info(self.h, o.ctypes.data_as(_ip), None)
        if o[0] < 0:
            return None
        gid = np.zeros(int(o[1]) * int(o[2]), dtype=np.int32)
        lib().mmgh_mg_gather_info(self.h, o.ctypes.data_as(_ip), gid.ctypes.data_as(_ip))
        return int(o[0]), int(o[1]), int(o[2]), int(o[3]), gid.reshape(int(o[1]), int(o[2]))

    def level_part(self, l, nparts):
        part = np.zeros(self.grid(l).sizes()["n"], dtype=np.int32)
        lib().mmgh_mg_level_part(self.h, l, int(nparts), part.ctypes.data_as(_ip))
        return part

    def setup_exchange(self, rank, all_gather_object, exact=False):
        """Distributed run: register every level's ghost exchange with the device
        (mmg_level_set_exchange).  Needs mmg_comm_init to have been called.  exact: ghosts refreshed
        before every phase (mmg_level_set_exchange_mode) -- the V-cycle then reproduces the undecomposed
        (single-GPU / CPU reference) residual history; raises if some level admits no such schedule."""
        from . import _capi
        for l in range(self.nlevels):
            g = self.grid(l)
            if g.is_replicated():      # a complete copy on every rank: relaxed without any exchange
                continue
            no, gid, gown = g.local_map()
            nbr, sp, si, rp = build_exchange_lists(rank, no, gid, gown, all_gather_object)
            s = g.sizes()
            lv = _capi.Level.borrow(g.device_level(), s["n"], s["a_size"])
            lv.set_exchange(no, nbr, sp, si, rp)
            if exact:
                lv.set_exchange_mode(1)

    def setup_exchange_native(self, exact=False):
        """The C++ path (Multigrid::setup_exchange): lists worked out by extract_subdomain from the global
        hierarchy, no all_gather.  Needs mmg_comm_init."""
        _chk(lib().mmgh_mg_setup_exchange(self.h, int(exact)))

    @property
    def nlevels(self):
        return lib().mmgh_mg_nlevels(self.h)

    def grid(self, l):
        return Grid(lib().mmgh_mg_grid(self.h, l), owner=self)

    def transfer(self, which, l):
        """which: 'R' or 'P'; returns dict(rows, cols, colptr, rowidx, val) or None."""
        w = 0 if which == "R" else 1
        r, c, nnz = C.c_int(0), C.c_int(0), C.c_int(0)
        if lib().mmgh_mg_transfer_shape(self.h, w, l, C.byref(r), C.byref(c), C.byref(nnz)):
            return None
        colptr = np.zeros(c.value + 1, dtype=np.int32)
        rowidx = np.zeros(nnz.value, dtype=np.int32)
        val = np.zeros(nnz.value)
        lib().mmgh_mg_transfer_get(self.h, w, l, colptr.ctypes.data_as(_ip), rowidx.ctypes.data_as(_ip),
                                   val.ctypes.data_as(_dp))
        return dict(rows=r.value, cols=c.value, colptr=colptr, rowidx=rowidx, val=val)

    def vcycle(self):
        v = C.c_double(0)
        _chk(lib().mmgh_mg_vcycle(self.h, C.byref(v)))
        if v.value >= 0:
            self.residuals.append(v.value)
        return v.value

    def vcycles(self, n):
        res = np.zeros(n)
        ms = C.c_float(0)
        _chk(lib().mmgh_mg_vcycles(self.h, n, res.ctypes.data_as(_dp), C.byref(ms)))
        self.residuals.extend(res.tolist())
        return res, ms.value

    def residual(self):
        v = C.c_double(0)
        _chk(lib().mmgh_mg_residual(self.h, C.byref(v)))
        return v.value


class FracStepMultigrid(Multigrid):
    """run_fracstep_param's hierarchy (FractionalStepSim.cpp:115-121): a FractionalStepMultigrid over
    FractionalStepGrids; `step()` is one device-resident time step (mmg_fracstep_step)."""

    def __init__(self, clouds, polydegs, dim=2, dt=2e-4, mu=0.025, rho=1.0, ordering=ORDER_MC, tile_points=0):
        npts = _i([len(c) for c in clouds])
        xyz = _d(np.concatenate([_d(c).reshape(-1, 3) for c in clouds], axis=0))
        pd = _i(polydegs)
        self.h = lib().mmgh_mg_create_fs(len(clouds), npts.ctypes.data_as(_ip), xyz.ctypes.data_as(_dp),
                                         pd.ctypes.data_as(_ip), int(dim), dt, mu, rho, ordering, tile_points)
        if not self.h:
            raise HostError(_err())
        self.omega, self.iters, self.frac_step, self.residuals = 1.4, 5, True, []
        self.dt, self.mu, self.rho, self.dim = dt, mu, rho, dim

    def fs_grid(self):
        g = FracStepGrid(lib().mmgh_mg_grid(self.h, self.nlevels - 1), owner=self)
        g.dt, g.mu, g.rho, g.dim = self.dt, self.mu, self.rho, self.dim
        return g

    def step(self, max_cycles=1000):
        """-> (fs_residual, V-cycles taken)"""
        nc, r = C.c_int(0), C.c_double(0)
        _chk(lib().mmgh_mg_fs_step(self.h, int(max_cycles), C.byref(nc), C.byref(r)))
        return r.value, nc.value


# ---- synthetic clouds (vectorised; seeds per SURVEY 8d) ------------------------------------
def square_cloud(nside, seed=12345, jitter=0.25):
    """nside^2 lattice on [0,1]^2, interior jittered by +-jitter*h, boundary coordinates
    exactly 0/1 (the reference detects boundaries by exact compares, testing_functions.cpp:86)."""
    return box_cloud(nside, 2, seed, jitter)


# ---- reference-shaped ("Gmsh-like") clouds ---------------------------------------------------
# The reference's inputs are Gmsh triangulations (testing_functions.cpp:355-364): evenly spaced nodes on the boundary
# curves, a quasi-uniform interior whose first layer sits about one triangle height (0.87 h) off the boundary, and a
# node on the bisector of every corner.  The generators below reproduce those three properties deterministically in
# O(N); DESIGN section 2 shows why they matter (the one-sided Neumann stencils plus the implicit elimination are
# stable on such clouds for polyDeg 3-6 and not on lattices / clipped packings).
def _hex_rows(lo, hi, h):
    """Hexagonally packed points of the square [lo, hi]^2: an ODD number of rows (spacing ~ 0.87 h), full rows
    lo..hi alternating with rows shifted by half a spacing -- first and last row are full rows, so each of the four
    corners (lo|hi, lo|hi) holds a point: the cloud is symmetric about the diagonals of the square's corners."""
    rows = max(3, int(round((hi - lo) / (h * math.sqrt(3.0) / 2.0))) + 1)
    rows += 1 - rows % 2
    m = max(2, int(round((hi - lo) / h)) + 1)
    ys = lo + (hi - lo) * np.arange(rows) / (rows - 1)
    xe = lo + (hi - lo) * np.arange(m) / (m - 1)
    xo = lo + (hi - lo) * (np.arange(m - 1) + 0.5) / (m - 1)
    ne, no = (rows + 1) // 2, rows // 2
    X = np.concatenate([np.tile(xe, ne), np.tile(xo, no)])
    Y = np.concatenate([np.repeat(ys[0::2], m), np.repeat(ys[1::2], m - 1)])
    return X, Y


def _square_boundary(nside):
    """4 (nside - 1) evenly spaced nodes on the unit square, coordinates exactly 0 / 1 on the sides
    (the reference's boundary test is `x==0||x==1||y==0||y==1`, testing_functions.cpp:86,180)."""
    t = np.arange(nside) * (1.0 / (nside - 1))
    t[-1] = 1.0
    z, o = np.zeros(nside), np.ones(nside)
    bx = np.concatenate([t, t, z[1:-1], o[1:-1]])
    by = np.concatenate([z, o, t[1:-1], t[1:-1]])
    return bx, by


def quasi_uniform_square_cloud(nside, offset=0.8):
    """Gmsh-like cloud of the unit square with boundary spacing h = 1 / (nside - 1): evenly spaced boundary nodes,
    hexagonally packed interior kept `offset` h off the boundary, a node on every corner bisector.  About
    1.155 (nside - 2)^2 + 4 (nside - 1) points.  Boundary nodes first."""
    h = 1.0 / (nside - 1)
    bx, by = _square_boundary(nside)
    X, Y = _hex_rows(offset * h, 1.0 - offset * h, h)
    x, y = np.concatenate([bx, X]), np.concatenate([by, Y])
    return np.stack([x, y, np.zeros(len(x))], axis=1)


def _ring(r0, m, phase=0.0):
    th = 2.0 * np.pi * (np.arange(m) + phase) / m
    return 0.5 + r0 * np.cos(th), 0.5 + r0 * np.sin(th)


def quasi_uniform_square_with_circle_cloud(nside, offset=0.8):
    """Gmsh-like cloud of the reference's "square_with_circle" geometry (unit square minus the disc of radius 0.25
    around (0.5, 0.5), testing_functions.cpp:85-106,186-209): evenly spaced nodes on the square and ON the circle
    (|r^2 - 1/16| <= 1e-10), one conforming ring of interior nodes `offset` h outside the circle (staggered against
    the circle's nodes), hexagonal packing beyond it."""
    h = 1.0 / (nside - 1)
    bx, by = _square_boundary(nside)
    m = max(8, int(round(2 * np.pi * 0.25 / h)))
    cx, cy = _ring(0.25, m)
    r1 = 0.25 + offset * h
    lx, ly = _ring(r1, max(8, int(round(2 * np.pi * r1 / h))), 0.5)
    X, Y = _hex_rows(offset * h, 1.0 - offset * h, h)
    keep = np.sqrt((X - 0.5) ** 2 + (Y - 0.5) ** 2) >= r1 + 0.75 * h
    x = np.concatenate([bx, cx, lx, X[keep]])
    y = np.concatenate([by, cy, ly, Y[keep]])
    return np.stack([x, y, np.zeros(len(x))], axis=1)


def quasi_uniform_annulus_cloud(nr):
    """Gmsh-like cloud of the reference's "concentric_circles" geometry (0.25 <= r <= 0.5 around (0.5, 0.5),
    testing_functions.cpp:107-135,212-250): nr + 1 concentric rings, radial spacing 0.25 / nr = one triangle height,
    node spacing h = that / 0.866 along every ring, consecutive rings staggered; inner and outer ring ON the circles."""
    hr = 0.25 / nr
    h = hr / (math.sqrt(3.0) / 2.0)
    xs, ys = [], []
    for i in range(nr + 1):
        r0 = 0.25 + i * hr if i < nr else 0.5
        x, y = _ring(r0, max(8, int(round(2 * np.pi * r0 / h))), 0.5 * (i % 2))
        xs.append(x)
        ys.append(y)
    x, y = np.concatenate(xs), np.concatenate(ys)
    return np.stack([x, y, np.zeros(len(x))], axis=1)


def square_with_circle_cloud(nside, seed=12345, jitter=0.25):
    """Unit square with a circular hole of radius 0.25 around (0.5, 0.5): the jittered lattice of square_cloud without
    the points inside (or within 0.6 h of) the circle, plus a ring of points ON the circle (|r^2 - 1/16| <= 1e-10)."""
    pts = box_cloud(nside, 2, seed, jitter)
    h = 1.0 / (nside - 1)
    r = np.sqrt((pts[:, 0] - 0.5) ** 2 + (pts[:, 1] - 0.5) ** 2)
    pts = pts[r >= 0.25 + 0.6 * h]
    m = max(8, int(round(2 * np.pi * 0.25 / h)))
    th = 2 * np.pi * np.arange(m) / m
    ring = np.stack([0.5 + 0.25 * np.cos(th), 0.5 + 0.25 * np.sin(th), np.zeros(m)], axis=1)
    return np.concatenate([pts, ring], axis=0)


def annulus_cloud(nr, seed=12345, jitter=0.25):
    """Point cloud of the annulus 0.25 <= r <= 0.5 around (0.5, 0.5): nr + 1 rings, spacing h = 0.25 / nr along r and
    about h along each ring; the inner and outer ring lie ON the circles (the reference detects them by
    |r^2 - R^2| <= 1e-10, testing_functions.cpp:124,129), the others are jittered by +-jitter*h in r and along the ring."""
    rng = np.random.default_rng(seed)
    h = 0.25 / nr
    pts = []
    for i in range(nr + 1):
        r0 = 0.25 + i * h
        m = max(8, int(round(2 * np.pi * r0 / h)))
        th = 2 * np.pi * (np.arange(m) + 0.5 * (i % 2)) / m
        r = np.full(m, r0)
        if 0 < i < nr:
            r = r + jitter * h * rng.uniform(-1, 1, m)
            th = th + jitter * h / r0 * rng.uniform(-1, 1, m)
        pts.append(np.stack([0.5 + r * np.cos(th), 0.5 + r * np.sin(th), np.zeros(m)], axis=1))
    return np.concatenate(pts, axis=0)


def box_cloud(nside, dim, seed=12345, jitter=0.25, edges=True):
    """nside^dim lattice on the unit box, interior jittered by +-jitter*h, boundary coordinates exactly 0/1.
    edges=False (3-D): no nodes on the edges and corners of the box -- every boundary node then lies on exactly one
    face and has ONE normal; the one-sided n.grad stencils of a Neumann problem are badly conditioned on edge nodes
    (|a_ii| / sum|a_ij| = 0.04-0.1 against 0.23-0.5 on the faces) and the V-cycle diverges with them."""
    rng = np.random.default_rng(seed)
    h = 1.0 / (nside - 1)
    ax = np.arange(nside) * h
    ax[-1] = 1.0
    if dim == 2:
        Y, X = np.meshgrid(ax, ax, indexing="ij")
        pts = np.stack([X.ravel(), Y.ravel(), np.zeros(nside * nside)], axis=1)
        idx = np.stack(np.meshgrid(np.arange(nside), np.arange(nside), indexing="ij"), axis=-1).reshape(-1, 2)
    else:
        Z, Y, X = np.meshgrid(ax, ax, ax, indexing="ij")
        pts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        idx = np.stack(np.meshgrid(np.arange(nside), np.arange(nside), np.arange(nside), indexing="ij"), axis=-1).reshape(-1, 3)
    interior = np.all((idx > 0) & (idx < nside - 1), axis=1)
    jit = (rng.random((len(pts), dim)) * 2 - 1) * jitter * h
    pts[interior, :dim] += jit[interior]
    if not edges and dim == 3:
        faces = ((idx == 0) | (idx == nside - 1)).sum(axis=1)
        pts = pts[faces <= 1]
    return pts


# ---- domain decomposition helpers (setup-time, host side) --------------------------------
def build_exchange_lists(rank, n_owned, gid, ghost_owner, all_gather_object):
    """Who sends what to whom.  Ghosts are grouped by owner (ascending), so the values
    received from one neighbour form one contiguous segment of the ghost range.
    all_gather_object(obj) -> list of every rank's obj (torch.distributed / a test stub).
    Returns (nbr_rank, send_ptr, send_idx, recv_ptr) for mmg_level_set_exchange."""
    ghost_gid = gid[n_owned:]
    need = {}
    for o in np.unique(ghost_owner):
        need[int(o)] = ghost_gid[ghost_owner == o]
    everyone = all_gather_object(need)           # everyone[q][r] = gids q needs from r
    order = np.argsort(gid[:n_owned], kind="stable")
    sorted_gid = gid[:n_owned][order]
    nbrs = sorted(set(need.keys()) | {q for q, d in enumerate(everyone) if rank in d and q != rank})
    send_ptr, send_idx, recv_ptr = [0], [], [0]
    for q in nbrs:
        wanted = everyone[q].get(rank, np.zeros(0, dtype=np.int32)) if q != rank else np.zeros(0, dtype=np.int32)
        pos = np.searchsorted(sorted_gid, wanted)
        assert np.all(sorted_gid[pos] == wanted), "a neighbour asks for a point this rank does not own"
        send_idx.extend(order[pos].tolist())
        send_ptr.append(len(send_idx))
        recv_ptr.append(recv_ptr[-1] + len(need.get(q, [])))
    return (np.array(nbrs, dtype=np.int32), np.array(send_ptr, dtype=np.int32), np.array(send_idx, dtype=np.int32),
            np.array(recv_ptr, dtype=np.int32))


def _jitter_hash(gid, axis, seed):
    """Counter-based uniform(-1,1): every rank computes the same jitter for a lattice point."""
    z = (gid.astype(np.uint64) * np.uint64(3) + np.uint64(axis)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def slab_bounds(rank, nranks, nx_glob):
    """x-layers [lo, hi) of rank's slab when nx_glob layers are shared as evenly as possible."""
    base, rem = divmod(nx_glob, nranks)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def slab_cloud(rank, nranks, nside, dim=3, margin=5, seed=12345, jitter=0.25, total=False):
    """Rank's part of a jittered lattice cut into x-slabs: its own x-layers plus `margin` layers of each
    neighbour as ghost candidates.  total = False (weak scaling): a (nranks*nside) x nside [x nside] lattice on
    [0,nranks]x[0,1]^(dim-1), nside layers per rank.  total = True (strong scaling): ONE nside^dim lattice on the
    unit cube, its nside layers shared evenly (slab_bounds).
    Returns (points, flags, gid, owner); flags 0 interior / 1 global Dirichlet boundary / 3 margin."""
    h = 1.0 / (nside - 1)
    nx_glob = nside if total else nranks * nside
    own_lo, own_hi = slab_bounds(rank, nranks, nx_glob)
    lo = max(0, own_lo - margin)
    hi = min(nx_glob, own_hi + margin)
    ix = np.arange(lo, hi)
    if dim == 3:
        IZ, IY, IX = np.meshgrid(np.arange(nside), np.arange(nside), ix, indexing="ij")
        idx = np.stack([IX.ravel(), IY.ravel(), IZ.ravel()], axis=1)
        gid = (idx[:, 2] * nside + idx[:, 1]) * nx_glob + idx[:, 0]
    else:
        IY, IX = np.meshgrid(np.arange(nside), ix, indexing="ij")
        idx = np.stack([IX.ravel(), IY.ravel(), np.zeros(IX.size, dtype=np.int64)], axis=1)
        gid = idx[:, 1] * nx_glob + idx[:, 0]
    pts = idx.astype(np.float64) * h
    bnd = (idx[:, 0] == 0) | (idx[:, 0] == nx_glob - 1) | (idx[:, 1] == 0) | (idx[:, 1] == nside - 1)
    if dim == 3:
        bnd |= (idx[:, 2] == 0) | (idx[:, 2] == nside - 1)
    for a in range(dim):
        pts[~bnd, a] += _jitter_hash(gid[~bnd], a, seed) * jitter * h
    starts = np.array([slab_bounds(r, nranks, nx_glob)[0] for r in range(nranks)])
    owner = (np.searchsorted(starts, idx[:, 0], side="right") - 1).astype(np.int32)
    flags = np.where(owner == rank, bnd.astype(np.int32), 3).astype(np.int32)
    return pts, flags, gid.astype(np.int32), owner


def block_dims(nranks, dim=3):
    """Ranks per axis of the box decomposition: as cubic as possible (8 -> 2 x 2 x 2, 4 -> 2 x 2 x 1, 6 -> 3 x 2 x 1,
    a prime count -> slabs)."""
    dims = [1] * 3
    n = int(nranks)
    f = 2
    factors = []
    while n > 1:
        while n % f == 0:
            factors.append(f)
            n //= f
        f += 1
    for f in sorted(factors, reverse=True):
        a = min(range(dim), key=lambda k: dims[k])
        dims[a] *= f
    dims[:dim] = sorted(dims[:dim], reverse=True)
    return tuple(dims)


def block_cloud(rank, nranks, nside, dim=3, margin=5, seed=12345, jitter=0.25, total=False):
    """Rank's part of a jittered lattice cut into BOXES (configs[3] names 2 x 2 x 2 sub-domains): its own
    block of lattice points plus `margin` layers around it (faces, edges and corners) as ghost candidates.  Ranks per
    axis from block_dims.  total = False (weak scaling): every rank owns nside^dim points of a (px nside) x (py nside)
    x (pz nside) lattice; total = True (strong scaling): ONE nside^dim lattice on the unit cube shared by the boxes.
    Same return convention and the same jitter (a hash of the global id) as slab_cloud."""
    p = block_dims(nranks, dim)
    nglob = [nside if total else p[a] * nside for a in range(3)]
    if dim == 2:
        nglob[2] = 1
    h = 1.0 / (nside - 1)
    b = [rank % p[0], (rank // p[0]) % p[1], rank // (p[0] * p[1])]
    rng_ax, own = [], []
    for a in range(3):
        if a >= dim:
            rng_ax.append(np.arange(1))
            own.append((0, 1))
            continue
        lo, hi = slab_bounds(b[a], p[a], nglob[a])
        own.append((lo, hi))
        rng_ax.append(np.arange(max(0, lo - margin), min(nglob[a], hi + margin)))
    IZ, IY, IX = np.meshgrid(rng_ax[2], rng_ax[1], rng_ax[0], indexing="ij")
    idx = np.stack([IX.ravel(), IY.ravel(), IZ.ravel()], axis=1)
    gid = (idx[:, 2] * nglob[1] + idx[:, 1]) * nglob[0] + idx[:, 0]
    pts = idx.astype(np.float64) * h
    bnd = np.zeros(len(idx), dtype=bool)
    for a in range(dim):
        bnd |= (idx[:, a] == 0) | (idx[:, a] == nglob[a] - 1)
    for a in range(dim):
        pts[~bnd, a] += _jitter_hash(gid[~bnd], a, seed) * jitter * h
    owner = np.zeros(len(idx), dtype=np.int64)
    mult = [1, p[0], p[0] * p[1]]
    for a in range(dim):
        starts = np.array([slab_bounds(r, p[a], nglob[a])[0] for r in range(p[a])])
        owner += mult[a] * (np.searchsorted(starts, idx[:, a], side="right") - 1)
    owner = owner.astype(np.int32)
    flags = np.where(owner == rank, bnd.astype(np.int32), 3).astype(np.int32)
    return pts, flags, gid.astype(np.int32), owner
