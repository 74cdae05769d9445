"""Shared test helpers: fixture loading, oracle/device/emulator construction."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SUPPORT = os.path.join(ROOT, "tests", "support")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def ensure_built():
    from oracle import oracle_c
    if not os.path.exists(os.path.join(ROOT, "oracle", "libmmg_oracle.so")):
        oracle_c.build()
    emu = os.path.join(SUPPORT, "libplan_emulate.so")
    srcs = [os.path.join(SUPPORT, "plan_emulate.cpp"),
            os.path.join(ROOT, "meshlessmultigridpoisson_amd", "csrc", "device", "plan.cpp"),
            os.path.join(ROOT, "meshlessmultigridpoisson_amd", "csrc", "device", "level_plan.cpp")]
    if not os.path.exists(emu) or any(os.path.getmtime(s) > os.path.getmtime(emu) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", emu] + srcs, check=True)


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    case = {k: z[k] for k in z.files}
    case["nlevels"] = int(case["nlevels"])
    return case


def level_arrays(case, i):
    p = f"L{i}_"
    n, a_size, neumann, iters = [int(v) for v in case[p + "meta"]]
    return dict(n=n, rowptr=case[p + "rowptr"], col=case[p + "col"], val=case[p + "val"],
                bcflags=case[p + "bcflags"], neumann=neumann, omega=float(case[p + "omega"]), iters=iters,
                btype=case[p + "btype"], bptr=case[p + "bptr"], bpts=case[p + "bpts"], bvals=case[p + "bvals"],
                x0=case[p + "x0"], b0=case[p + "b0"], a_size=a_size)


def oracle_level(la):
    from oracle import oracle_c as oc
    return oc.Level(la["n"], la["rowptr"], la["col"], la["val"], la["x0"], la["b0"], la["bcflags"], la["neumann"],
                    la["omega"], la["iters"], la["btype"], la["bptr"], la["bpts"], la["bvals"])


def oracle_multigrid(case):
    from oracle import oracle_c as oc
    nl = case["nlevels"]
    levels = [oracle_level(level_arrays(case, i)) for i in range(nl)]
    R, P = [None] * nl, [None] * nl
    for i in range(nl):
        if f"R{i}_shape" in case:
            R[i] = oc.Transfer(*case[f"R{i}_shape"], case[f"R{i}_colptr"], case[f"R{i}_rowidx"], case[f"R{i}_val"])
        if f"P{i}_shape" in case:
            P[i] = oc.Transfer(*case[f"P{i}_shape"], case[f"P{i}_colptr"], case[f"P{i}_rowidx"], case[f"P{i}_val"])
    return oc.Multigrid(levels, R, P, frac_step=bool(case["frac_step"]))


def device_level(la, **kw):
    from meshlessmultigridpoisson_amd import _capi
    return _capi.Level(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], la["neumann"], la["omega"],
                       la["iters"], la["btype"], la["bptr"], la["bpts"], la["bvals"], x=la["x0"], b=la["b0"], **kw)


def device_hierarchy(case, frac_step=None, **kw):
    from meshlessmultigridpoisson_amd import _capi
    nl = case["nlevels"]
    levels = [device_level(level_arrays(case, i), **kw) for i in range(nl)]
    R, P = [None] * nl, [None] * nl
    for i in range(nl):
        if f"R{i}_shape" in case:
            R[i] = _capi.Transfer(*case[f"R{i}_shape"], case[f"R{i}_colptr"], case[f"R{i}_rowidx"], case[f"R{i}_val"])
        if f"P{i}_shape" in case:
            P[i] = _capi.Transfer(*case[f"P{i}_shape"], case[f"P{i}_colptr"], case[f"P{i}_rowidx"], case[f"P{i}_val"])
    fs = bool(case["frac_step"]) if frac_step is None else frac_step
    return _capi.Hierarchy(levels, R, P, frac_step=fs)


def oracle_of_multigrid(mg):
    """The hierarchy a host `Multigrid` handle built, as CPU-oracle objects (checker only)."""
    from oracle import oracle_c as oc
    nl = mg.nlevels
    levels = []
    for l in range(nl):
        g = mg.grid(l)
        w, it = g.relaxation() if hasattr(g, "relaxation") else (mg.omega, mg.iters)   # per grid (gridclasses.hpp:6-14)
        la = g.level_arrays(w, it)
        levels.append(oc.Level(la["n"], la["rowptr"], la["col"], la["val"], la["x0"], la["b0"], la["bcflags"],
                               la["neumann"], la["omega"], la["iters"], la["btype"], la["bptr"], la["bpts"],
                               la["bvals"]))
    R, P = [None] * nl, [None] * nl
    for l in range(nl):
        t = mg.transfer("R", l)
        if t:
            R[l] = oc.Transfer.from_dict(t)
        t = mg.transfer("P", l)
        if t:
            P[l] = oc.Transfer.from_dict(t)
    om = oc.Multigrid(levels, R, P, frac_step=bool(getattr(mg, "frac_step", False)))
    om.damping = float(getattr(mg, "damping", 1.0))     # Multigrid.set_correction_damping (opt-in, not in the reference)
    return om


def oracle_of_fracstep(g):
    """Oracle objects over the operators a host `FracStepGrid` built (checker only)."""
    from oracle import oracle_c as oc
    nx, ny = g.normals()
    _bt, _bp, bpts, _bv = g.boundaries()
    if getattr(g, "dim", 2) >= 3:
        return oc.FracStep3(g.sizes()["n"], g.op(0), g.op(1), g.op(3), g.op(2), nx, ny, g.normal_z(), bpts)
    return oc.FracStep(g.sizes()["n"], g.op(0), g.op(1), g.op(2), nx, ny, bpts)


def oracle_fracstep_time_step(om, ofs, g_arrays, dt, mu, rho, tol, max_cycles, vcycle=None):
    """One time step of run_fracstep_param (FractionalStepSim.cpp:131-147) with oracle objects: om the oracle
    Multigrid (frac_step), ofs the oracle FracStep(3) of its finest grid, g_arrays = dict(bpts, bvals (list of per-
    component boundary value arrays), coupling=((rp, col, val), diag), bcflags).  Returns (fs_residual, cycles)."""
    from oracle import oracle_c as oc
    fine = om.levels[-1]
    n = ofs.n
    comps = [ofs.u, ofs.v] + ([ofs.w] if hasattr(ofs, "w") else [])
    for c, vals in zip(comps, g_arrays["bvals"]):          # set_uv_bound
        c[g_arrays["bpts"]] = vals
    ofs.calc_hat(dt, mu, rho)
    ofs.set_ppe_source(fine.b, dt, rho)
    oc.push_inhomog(n, g_arrays["coupling"][0], g_arrays["coupling"][1], g_arrays["bcflags"], fine.b)
    cycles = 0
    while om.residual() >= tol and cycles < max_cycles:
        (vcycle or om.vcycle)()          # vcycle: e.g. the multi-GPU relaxation schedule, om.vcycle_hybrid(parts, n)
        fine.bound_eval_neumann()
        cycles += 1
    ofs.correct(fine.x[:n], dt, rho)
    for c, vals in zip(comps, g_arrays["bvals"]):
        c[g_arrays["bpts"]] = vals
    return ofs.residual(), cycles


# ---- CPU interpreter of the packed plan (tests/support/plan_emulate.cpp) ------------
_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        ensure_built()
        L = C.CDLL(os.path.join(SUPPORT, "libplan_emulate.so"))
        L.emu_level_create.restype = C.c_void_p
        L.emu_last_error.restype = C.c_char_p
        L.emu_level_destroy.argtypes = [C.c_void_p]
        L.emu_level_info.argtypes = [C.c_void_p, _ip]
        L.emu_level_sweeps.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_int]
        L.emu_level_bound_eval.argtypes = [C.c_void_p, _dp, _dp]
        L.emu_level_residual.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.emu_level_residual.restype = C.c_double
        L.emu_level_sor_phases.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.emu_level_owned_sum.argtypes = [C.c_void_p, _dp]
        L.emu_level_sor_one_phase.argtypes = [C.c_void_p, _dp, _dp, C.c_double, C.c_int]
        L.emu_level_point_phases.argtypes = [C.c_void_p, C.c_void_p, _ip, C.POINTER(C.c_ulonglong)]
        L.emu_level_owned_sum.restype = C.c_double
        L.emu_level_slot_bits.argtypes = [C.c_void_p]
        L.emu_level_waves.argtypes = [C.c_void_p]
        L.emu_level_dense_long.argtypes = [C.c_void_p]
        L.emu_level_stream_bytes.argtypes = [C.c_void_p]
        L.emu_level_stream_bytes.restype = C.c_longlong
        L.emu_level_nnz.argtypes = [C.c_void_p]
        L.emu_level_nnz.restype = C.c_longlong
        L.emu_transfer_apply.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp, C.c_int, C.c_int]
        _emu = L
    return _emu


class EmuLevel:
    def __init__(self, la, tile_ptr=None, tile_size=0, lanes_per_row=0, tile_phase=None, waves_per_tile=0):
        from meshlessmultigridpoisson_amd import _capi
        d, self._keep = _capi.make_desc(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], la["neumann"],
                                        la["omega"], la["iters"], la["btype"], la["bptr"], la["bpts"], la["bvals"],
                                        tile_ptr, tile_size, lanes_per_row, tile_phase, waves_per_tile)
        self.L = emu_lib()
        self._desc = d
        self._neumann = bool(la["neumann"])
        self.h = self.L.emu_level_create(C.byref(d))
        if not self.h:
            raise RuntimeError(self.L.emu_last_error().decode())
        self.x = np.array(la["x0"], dtype=np.float64)
        self.b = np.array(la["b0"], dtype=np.float64)
        self.omega = la["omega"]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.emu_level_destroy(self.h)

    def info(self):
        out = np.zeros(6, dtype=np.int32)
        self.L.emu_level_info(self.h, out.ctypes.data_as(_ip))
        return dict(zip(["n_tiles", "n_phases", "n_groups", "max_slots", "b_tiles", "b_phases"], out.tolist()))

    def waves(self):
        return int(self.L.emu_level_waves(self.h))

    def dense_long(self):
        return bool(self.L.emu_level_dense_long(self.h))

    def sweeps(self, k):
        self.L.emu_level_sweeps(self.h, self.x.ctypes.data_as(_dp), self.b.ctypes.data_as(_dp), self.omega, int(k))

    def sor_phases(self):
        self.L.emu_level_sor_phases(self.h, self.x.ctypes.data_as(_dp), self.b.ctypes.data_as(_dp), self.omega)

    def sor_one_phase(self, ph):
        return self.L.emu_level_sor_one_phase(self.h, self.x.ctypes.data_as(_dp), self.b.ctypes.data_as(_dp),
                                              C.c_double(self.omega), int(ph))

    def point_phases(self):
        n = len(self.x) - (1 if self._neumann else 0)
        ph = np.zeros(n, dtype=np.int32)
        gm = np.zeros(n, dtype=np.uint64)
        rc = self.L.emu_level_point_phases(self.h, C.byref(self._desc), ph.ctypes.data_as(_ip),
                                           gm.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        assert rc == 0, self.L.emu_last_error()
        return ph, gm

    def owned_sum(self):
        return float(self.L.emu_level_owned_sum(self.h, self.x.ctypes.data_as(_dp)))

    def bound_eval(self):
        self.L.emu_level_bound_eval(self.h, self.x.ctypes.data_as(_dp), self.b.ctypes.data_as(_dp))

    def residual(self):
        r = np.zeros_like(self.x)
        nrm = self.L.emu_level_residual(self.h, self.x.ctypes.data_as(_dp), self.b.ctypes.data_as(_dp),
                                        r.ctypes.data_as(_dp))
        return r, nrm


def emu_transfer_apply(shape, colptr, rowidx, val, x, add_to=None, L=4):
    rows, cols = int(shape[0]), int(shape[1])
    colptr = np.ascontiguousarray(colptr, dtype=np.int32)
    rowidx = np.ascontiguousarray(rowidx, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(rows) if add_to is None else np.array(add_to, dtype=np.float64)
    rc = emu_lib().emu_transfer_apply(rows, cols, colptr.ctypes.data_as(_ip), rowidx.ctypes.data_as(_ip),
                                      val.ctypes.data_as(_dp), x.ctypes.data_as(_dp), y.ctypes.data_as(_dp),
                                      int(add_to is not None), int(L))
    assert rc == 0, emu_lib().emu_last_error()
    return y


def rho_evaluation_noise(lv):
    """First-order bound of the rounding error of rho = ||b - A x||_1 / ||b||_1 evaluated in ANY association order:
    eps * || |A||x| + |b| ||_1 / ||b||_1 for the oracle Level `lv` in its current state.  Two correct evaluations
    (the CPU's sequential row sums, the GPU's lane-parallel ones) may differ by this much whatever rho is; it grows
    with 1/h^2 and with the weight size of high-degree stencils (2e-13 on the 600-point fixtures, 4e-12 on a
    10874-point polyDeg-6 Neumann level)."""
    import scipy.sparse as sp
    A = sp.csr_matrix((np.abs(lv.val), lv.col, lv.rowptr), shape=(lv.a_size, lv.a_size))
    return float(np.finfo(np.float64).eps * ((A @ np.abs(lv.x)).sum() + np.abs(lv.b).sum()) / np.abs(lv.b).sum())


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def ragged_level(n, seed, max_len=40, dirichlet_frac=0.0):
    """A level outside anything a stencil generator produces: random diagonally dominant CSR with RAGGED rows (1 ..
    max_len stored entries, the 1-entry rows holding the diagonal only), optionally a share of Dirichlet points."""
    rng = np.random.default_rng(seed)
    rowptr, col, val = [0], [], []
    flags = (rng.random(n) < dirichlet_frac).astype(np.int32)
    for i in range(n):
        k = int(rng.integers(1, max(2, min(max_len, n)) + 1))
        others = (rng.choice(np.delete(np.arange(n), i), size=min(k - 1, n - 1), replace=False) if n > 1
                  else np.zeros(0, dtype=np.int64))
        c = np.sort(np.append(others, i)).astype(np.int64)
        v = -rng.random(len(c))
        v[c == i] = 1.0 + len(c)
        col += c.tolist()
        val += v.tolist()
        rowptr.append(len(col))
    bpts = np.flatnonzero(flags).astype(np.int32)
    nb = 1 if len(bpts) else 0
    return dict(n=n, a_size=n, rowptr=np.array(rowptr, dtype=np.int32), col=np.array(col, dtype=np.int32), val=np.array(val),
                bcflags=flags, neumann=0, omega=1.3, iters=2, btype=np.full(nb, 1, dtype=np.int32),
                bptr=np.array([0, len(bpts)][: nb + 1], dtype=np.int32), bpts=bpts, bvals=rng.standard_normal(len(bpts)),
                x0=rng.standard_normal(n), b0=rng.standard_normal(n))
