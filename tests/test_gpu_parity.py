"""GPU parity tests proper: every call goes through the C-ABI of libmmgp.so
(hand-written gfx950 kernels) and is compared with the CPU oracle on identical,
identically ordered inputs.

Tolerances (fp64): single operations 1e-12 relative (only the association order
inside a row's dot product and in the norm reductions differs).  V-cycle residual
history: |rho_gpu - rho_cpu| <= 1e-10 * rho_cpu + FLOOR per cycle (BASELINE.json
north_star: 1e-10 relative).  FLOOR = 2e-13 is the fp64 evaluation noise of
rho = ||b - A x||_1 / ||b||_1 itself for these matrices (eps * |A||x| / |b| ~ 1e-13:
two correct CPU evaluations that associate a row's dot product differently
disagree by that much), so below rho ~ 1e-3 no implementation can agree to 1e-10
*relative*; DESIGN.md "Parity" discusses the window.
"""

FLOOR = 2e-13
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

CASES = ["dirichlet_3level", "neumann_2level", "neumann_3level", "dirichlet_2level_inhomog", "neumann_live_L6_3level"]


def _need_gpu():
    from meshlessmultigridpoisson_amd import _capi
    assert _capi.device_count() >= 1, "no HIP device visible: libmmgp has no CPU fallback"


@pytest.fixture(params=[4, 0], ids=["single-launch", "per-phase"])
def sweep_mode(request):
    """Both relaxation drivers: the dependency-driven single launch (default) and one
    launch per phase."""
    from meshlessmultigridpoisson_amd import _capi
    _capi.set_option("persistent_sweep", request.param)
    yield request.param
    _capi.set_option("persistent_sweep", 1)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tile,L", [(0, 0), (64, 2), (200, 8), (512, 1), (128, 16)])
def test_sweeps_residual_match_oracle(name, tile, L, sweep_mode):
    _need_gpu()
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    d = H.device_level(la, tile_size=tile, lanes_per_row=L)
    o.boundary_op(0)
    d.boundary_op(0)
    assert H.rel_err(d.get_x(), o.x) == 0.0
    o.sor_sweeps(1)
    d.sweeps(1)
    assert H.rel_err(d.get_x(), o.x) < 1e-12
    assert H.rel_err(d.get_x(), case["fine_x_after_1sweep"]) < 1e-12  # committed golden vector
    o.sor_sweeps(o.iters - 1)
    d.sweeps(o.iters - 1)
    assert H.rel_err(d.get_x(), o.x) < 1e-12
    assert H.rel_err(d.get_x(), case["fine_x_after_sor"]) < 1e-12
    r, ro = d.residual(), o.residual()
    assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(o.b).max())
    assert abs(d.residual_ratio() - o.residual_ratio()) <= 1e-10 * o.residual_ratio()
    assert abs(d.residual_ratio() - float(case["fine_ratio_after_sor"])) <= 1e-10 * float(case["fine_ratio_after_sor"])


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tile,L,waves", [(64, 8, 4), (128, 16, 4), (96, 8, 2), (200, 8, 6), (48, 16, 3), (128, 8, -1), (160, 16, -1)])
@pytest.mark.parametrize("mode", [0, 1, 4], ids=["per-phase", "auto", "single-launch"])
def test_dense_multiwave_kernels_match_oracle(name, tile, L, waves, mode):
    """Dense plans (mmg_level_desc.waves_per_tile > 1, or -1: ONE wavefront per tile): workgroups of `waves` wavefronts per tile, one barrier
    per round of mutually uncoupled rows (kernels_mw.hip) -- per-phase launches (tile_kernel_mw), the resident
    whole-sweep kernel ("auto" on these small levels: sweep_resident_mw) and the dependency-driven single launch
    (sweep_persistent_mw).  Same coupled-row order => the oracle's iterates to 1e-12; the three drivers agree bitwise."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    _capi.set_option("persistent_sweep", mode)
    try:
        d = H.device_level(la, tile_size=tile, lanes_per_row=L, waves_per_tile=waves)
        rowlen = int(np.diff(la["rowptr"])[:-1 if la["neumann"] else None][la["bcflags"] == 0].max())
        if waves == -1:   # one wavefront per tile: 8 lanes x <= 5 entries or 16 lanes x 3 entries, else the fallbacks below
            fits = rowlen - 2 <= (40 if L == 8 else 48)
        else:
            fits = rowlen - 2 <= 8 * L
        if fits:
            assert d.info()["waves_per_tile"] == waves
        else:   # the polyDeg-6 Neumann fixture: rows of ~190 entries take several row slots (4 / 6 wavefronts) or the packed stream
            assert d.info()["waves_per_tile"] in (1, 4, 6)
        o.boundary_op(0)
        d.boundary_op(0)
        o.sor_sweeps(1)
        d.sweeps(1)
        assert H.rel_err(d.get_x(), o.x) < 1e-12
        o.sor_sweeps(o.iters - 1)
        d.sweeps(o.iters - 1)
        assert H.rel_err(d.get_x(), o.x) < 1e-12
        r, ro = d.residual(), o.residual()
        assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(o.b).max())
        assert abs(d.residual_ratio() - o.residual_ratio()) <= 1e-10 * o.residual_ratio()
        x_mode = d.get_x()
        _capi.set_option("persistent_sweep", 0)
        ref = H.device_level(la, tile_size=tile, lanes_per_row=L, waves_per_tile=waves)
        ref.boundary_op(0)
        ref.sweeps(1)
        ref.sweeps(o.iters - 1)
        assert np.array_equal(ref.get_x(), x_mode)
    finally:
        _capi.set_option("persistent_sweep", 1)


@pytest.mark.parametrize("name", CASES)
def test_vcycle_residual_history_dense_levels(name):
    """Whole V-cycles with every level in the dense multi-wavefront layout (mmg_set_option("waves_per_tile", 4))."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    om = H.oracle_multigrid(case)
    _capi.set_option("waves_per_tile", 4)
    try:
        dh = H.device_hierarchy(case)
        assert all(l.info()["waves_per_tile"] == 4 for l in dh.levels)
        for k in range(len(case["resid_history"])):
            ro, rd = om.vcycle(), dh.vcycle()
            assert abs(rd - ro) <= 1e-10 * ro + max(FLOOR, H.rho_evaluation_noise(om.levels[-1])), (k, rd, ro)
        assert H.rel_err(dh.levels[-1].get_x(), om.levels[-1].x) < 1e-9
        # every coarser level too (round-2 review, weak 10).  A coarse level holds the CORRECTION of the last cycle, of
        # the size of the converged residual: the comparison is absolute, against the scale of the finest iterate --
        # 1e-9 * max|x_fine| (the relative form of this check failed at 2e-15 absolute on values of 1e-8).
        scale = np.abs(om.levels[-1].x).max()
        for lo, ld in zip(om.levels[:-1], dh.levels[:-1]):
            assert np.abs(ld.get_x() - lo.x).max() <= 1e-9 * scale
    finally:
        _capi.set_option("waves_per_tile", 0)


@pytest.mark.parametrize("name", ["neumann_2level", "neumann_3level"])
def test_bound_eval_and_rhs_ops(name):
    _need_gpu()
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    rng = np.random.default_rng(3)
    la["x0"] = rng.standard_normal(la["a_size"])
    la["b0"] = rng.standard_normal(la["a_size"])
    o = H.oracle_level(la)
    d = H.device_level(la)
    o.bound_eval_neumann()
    d.bound_eval_neumann()
    assert H.rel_err(d.get_x(), o.x) < 1e-12
    for coarse in (1, 0):
        o.modify_coeff_neumann(coarse)
        d.modify_coeff_neumann(coarse)
        assert np.array_equal(d.get_rhs(), o.b)
    d.zero_x()
    assert not d.get_x().any()


def test_boundary_op_inhomogeneous():
    _need_gpu()
    case = H.load_case("dirichlet_2level_inhomog")
    la = H.level_arrays(case, 1)
    o = H.oracle_level(la)
    d = H.device_level(la)
    for coarse in (0, 1, 0):
        o.boundary_op(coarse)
        d.boundary_op(coarse)
        assert np.array_equal(d.get_x(), o.x)
    newv = np.arange(len(la["bvals"]), dtype=np.float64)
    d.set_bvals(newv)
    d.boundary_op(0)
    x = d.get_x()
    assert np.array_equal(x[la["bpts"]], newv)


@pytest.mark.parametrize("name", ["dirichlet_3level", "neumann_2level", "neumann_3level"])
def test_restrict_prolong_match_oracle(name):
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    nl = case["nlevels"]
    om = H.oracle_multigrid(case)
    dh = H.device_hierarchy(case)
    fine_o, coarse_o = om.levels[nl - 1], om.levels[nl - 2]
    fine_d, coarse_d = dh.levels[nl - 1], dh.levels[nl - 2]
    fine_o.sor()
    fine_d.sor()
    # restriction (multigrid.cpp:81-86) restated with the oracle's pieces
    r = fine_o.residual()
    coarse_o.b[: coarse_o.n] = om.R[nl - 1].apply(r[: fine_o.n])
    from oracle import oracle_c as oc
    s = coarse_o.struct()
    oc.lib().orc_fix_vector_bound_coarse(s, coarse_o.b.ctypes.data_as(oc._dp))
    if fine_o.neumann:
        coarse_o.b[-1] = 0.0
        coarse_o.modify_coeff_neumann(1)
    _capi.restrict(fine_d, coarse_d, dh.R[nl - 1])
    scale = max(np.abs(coarse_o.b).max(), 1e-300)
    assert np.abs(coarse_d.get_rhs() - coarse_o.b).max() <= 1e-11 * scale
    # prolongation (multigrid.cpp:102-106)
    rng = np.random.default_rng(1)
    xc = rng.standard_normal(coarse_o.a_size)
    coarse_o.x[:] = xc
    coarse_d.set_x(xc)
    corr = om.P[nl - 2].apply(coarse_o.x[: coarse_o.n])
    if not fine_o.neumann:
        s = fine_o.struct()
        oc.lib().orc_fix_vector_bound_coarse(s, corr.ctypes.data_as(oc._dp))
    fine_o.x[: fine_o.n] += corr
    _capi.prolong_add(coarse_d, fine_d, dh.P[nl - 2])
    assert H.rel_err(fine_d.get_x(), fine_o.x) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_vcycle_residual_history(name, sweep_mode):
    """The headline parity gate: residual-per-V-cycle within 1e-10 relative of the
    CPU oracle AND of the committed golden history."""
    _need_gpu()
    case = H.load_case(name)
    om = H.oracle_multigrid(case)
    dh = H.device_hierarchy(case)
    gold = case["resid_history"]
    for k in range(len(gold)):
        ro = om.vcycle()
        rd = dh.vcycle()
        floor = max(FLOOR, H.rho_evaluation_noise(om.levels[-1]))   # 2e-13 on the small fixtures, ~1e-12 at polyDeg 6
        assert abs(rd - ro) <= 1e-10 * ro + floor, (k, rd, ro)
        assert abs(rd - gold[k]) <= 1e-10 * gold[k] + floor, (k, rd, gold[k])
    assert abs(dh.residual() - om.residual()) <= 1e-10 * om.residual() + 1e-13
    xo = om.levels[-1].x
    assert np.abs(dh.levels[-1].get_x() - xo).max() <= 1e-9 * np.abs(xo).max()


def test_vcycles_batched_and_fracstep_single_grid():
    _need_gpu()
    case = H.load_case("dirichlet_3level")
    dh = H.device_hierarchy(case)
    res, ms = dh.vcycles(5)
    assert np.allclose(res, case["resid_history"][:5], rtol=1e-10, atol=FLOOR)
    assert ms > 0
    # FracStepMultigrid.cpp:64-67: one grid -> vCycle is a bare sor(), nothing pushed
    from meshlessmultigridpoisson_amd import _capi
    la = H.level_arrays(H.load_case("neumann_2level"), 1)
    lv = H.device_level(la)
    h = _capi.Hierarchy([lv], [None], [None], frac_step=True)
    o = H.oracle_level(la)
    assert h.vcycle() == -1.0 and h.residuals == []
    o.sor()
    assert H.rel_err(lv.get_x(), o.x) < 1e-12


def test_generic_spmv():
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case("dirichlet_3level")
    la = H.level_arrays(case, 2)
    import scipy.sparse as sp
    A = sp.csr_matrix((la["val"], la["col"], la["rowptr"]), shape=(la["a_size"], la["a_size"]))
    x = np.random.default_rng(5).standard_normal(la["a_size"])
    m = _capi.Spmv(la["a_size"], la["a_size"], la["rowptr"], la["col"], la["val"])
    y = m.apply(x)
    ref = A @ x
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()


def test_rccl_ghost_exchange_single_rank_loopback():
    """The multi-GPU ghost refresh (mmg_level_exchange: pack kernel + grouped ncclSend/ncclRecv
    straight into the ghost segment) exercised on one GPU with the rank as its own neighbour:
    ghosts must become copies of the listed owned points, and a distributed sweep must equal
    'manual ghost copy + plain sweep'."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi, _host
    pts, flags, gid, owner = _host.slab_cloud(0, 2, 12, dim=3, margin=5)
    sub = _host.Grid.create_local(pts, flags, gid, owner, 3, 50, tile_points=256, lanes_per_row=2)
    n_owned, lgid, gown = sub.local_map()
    la = sub.level_arrays()
    n_ghost = la["n"] - n_owned
    assert n_ghost > 0
    rng = np.random.default_rng(9)
    x0 = rng.standard_normal(la["a_size"])
    b0 = rng.standard_normal(la["a_size"])
    b0[n_owned:] = 0.0
    send_idx = rng.integers(0, n_owned, size=n_ghost).astype(np.int32)

    def make():
        return _capi.Level(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], 0, 1.4, 5, la["btype"], la["bptr"],
                           la["bpts"], la["bvals"], x=x0, b=b0, tile_ptr=sub.tile_ptr(), lanes_per_row=2)

    _capi.comm_init(0, 1, _capi.comm_unique_id())
    try:
        d = make()
        d.set_exchange(n_owned, [0], [0, n_ghost], send_idx, [0, n_ghost])
        d.exchange()
        x = d.get_x()
        assert np.array_equal(x[n_owned:], x0[send_idx]) and np.array_equal(x[:n_owned], x0[:n_owned])
        d.sweeps(2)
        m = make()
        for _ in range(2):
            xm = m.get_x()
            xm[n_owned:] = xm[send_idx]
            m.set_x(xm)
            m.sweeps(1)
        assert np.array_equal(d.get_x(), m.get_x())
        # oracle cross-check of one local sweep with frozen ghosts
        o = H.oracle_level(dict(la, x0=x0, b0=b0))
        o.x[n_owned:] = o.x[send_idx]
        o.sor_sweeps(1)
        m2 = make()
        xm = m2.get_x()
        xm[n_owned:] = xm[send_idx]
        m2.set_x(xm)
        m2.sweeps(1)
        assert H.rel_err(m2.get_x(), o.x) < 1e-12
    finally:
        _capi.comm_finalize()


def test_per_phase_exchange_mode_single_rank_loopback():
    """mmg_level_set_exchange_mode(per_phase = 1) on one GPU with the rank as its own neighbour: the
    collective phase check, the lock-step phase loop with an RCCL ghost refresh before every phase.
    Expected iterates: the CPU interpreter of the same packed plan stepping phase by phase with manual
    ghost copies (tests/test_distributed_cpu.py proves that schedule equal to the sequential oracle).
    A ghost mapped to a point that its readers' phase also relaxes must be refused."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi, _host
    pts, flags, gid, owner = _host.slab_cloud(0, 2, 12, dim=3, margin=5)
    sub = _host.Grid.create_local(pts, flags, gid, owner, 3, 50, tile_points=256, lanes_per_row=2)
    n_owned, lgid, gown = sub.local_map()
    la = sub.level_arrays()
    n_ghost = la["n"] - n_owned
    rng = np.random.default_rng(9)
    la["x0"] = rng.standard_normal(la["a_size"])
    la["b0"] = rng.standard_normal(la["a_size"])
    la["b0"][n_owned:] = 0.0
    emu = H.EmuLevel(la, tile_ptr=sub.tile_ptr(), lanes_per_row=2, tile_phase=sub.tile_phase())
    ph, gm = emu.point_phases()
    nph = emu.info()["n_phases"]
    assert nph >= 4
    # a legal "neighbour": every ghost mirrors an owned interior point relaxed in a phase none of its readers runs in
    by_phase = [np.flatnonzero(ph[:n_owned] == p) for p in range(nph)]
    send_ok = np.zeros(n_ghost, dtype=np.int32)
    send_bad = np.zeros(n_ghost, dtype=np.int32)
    for j in range(n_ghost):
        mask = int(gm[n_owned + j])
        free = [p for p in range(nph) if not (mask >> p) & 1 and len(by_phase[p])]
        used = [p for p in range(nph) if (mask >> p) & 1 and len(by_phase[p])]
        send_ok[j] = rng.choice(by_phase[free[j % len(free)]])
        send_bad[j] = rng.choice(by_phase[used[0]]) if used else send_ok[j]
    assert any(int(gm[n_owned + j]) for j in range(n_ghost))

    def make():
        return _capi.Level(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], 0, 1.4, 5, la["btype"], la["bptr"],
                           la["bpts"], la["bvals"], x=la["x0"], b=la["b0"], tile_ptr=sub.tile_ptr(), lanes_per_row=2,
                           tile_phase=sub.tile_phase())

    _capi.comm_init(0, 1, _capi.comm_unique_id())
    try:
        d = make()
        assert np.array_equal(d.point_phases(), ph)
        d.set_exchange(n_owned, [0], [0, n_ghost], send_bad, [0, n_ghost])
        with pytest.raises(_capi.MmgError, match="no sequential order"):
            d.set_exchange_mode(1)
        d = make()
        d.set_exchange(n_owned, [0], [0, n_ghost], send_ok, [0, n_ghost])
        d.set_exchange_mode(1)
        d.sweeps(3)
        for _ in range(3):
            for p in range(nph):
                emu.x[n_owned:] = emu.x[send_ok]
                emu.sor_one_phase(p)
        xd = d.get_x()
        assert H.rel_err(xd[:n_owned], emu.x[:n_owned]) < 1e-12
        # the once-per-sweep schedule gives different iterates from the same state
        h = make()
        h.set_exchange(n_owned, [0], [0, n_ghost], send_ok, [0, n_ghost])
        h.sweeps(3)
        assert H.rel_err(h.get_x()[:n_owned], emu.x[:n_owned]) > 1e-8
    finally:
        _capi.comm_finalize()


@pytest.mark.parametrize("name", ["dirichlet_3level", "neumann_3level"])
@pytest.mark.parametrize("tile,L", [(64, 4), (32, 2), (200, 8)])
def test_persistent_single_launch_sweep_matches_oracle(name, tile, L):
    """mmg_set_option("persistent_sweep", 1): one launch per sweep, tiles started by their
    dependencies through agent-scope flags.  Same coupled-row order => same iterates."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    _capi.set_option("persistent_sweep", 4)
    try:
        d = H.device_level(la, tile_size=tile, lanes_per_row=L)
        assert d.info()["n_phases"] > 1
        o.boundary_op(0)
        d.boundary_op(0)
        for _ in range(3):
            o.sor_sweeps(2)
            d.sweeps(2)
            assert H.rel_err(d.get_x(), o.x) < 1e-12
        assert abs(d.residual_ratio() - o.residual_ratio()) <= 1e-10 * o.residual_ratio()
    finally:
        _capi.set_option("persistent_sweep", 1)


def test_persistent_sweep_3d_many_tiles():
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi, _host
    pts = _host.box_cloud(40, 3, seed=3)
    g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC, tile_points=128,
                                 lanes_per_row=2)
    la = g.level_arrays()
    rng = np.random.default_rng(2)
    la["b0"] = rng.standard_normal(la["a_size"])
    o = H.oracle_level(la)
    _capi.set_option("persistent_sweep", 4)
    try:
        d = _capi.Level(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], 0, 1.4, 5, la["btype"], la["bptr"],
                        la["bpts"], la["bvals"], x=la["x0"], b=la["b0"], tile_ptr=g.tile_ptr(), lanes_per_row=2)
        assert d.info()["n_tiles"] > 400
        o.sor_sweeps(4)
        d.sweeps(4)
        assert H.rel_err(d.get_x(), o.x) < 1e-12
        # fenced variant (full agent-scope acquire/release per tile) gives the same bits
        _capi.set_option("persistent_sweep", 2)
        x1 = d.get_x()
        d.sweeps(1)
        o.sor_sweeps(1)
        assert H.rel_err(d.get_x(), o.x) < 1e-12 and not np.array_equal(x1, d.get_x())
    finally:
        _capi.set_option("persistent_sweep", 1)


@pytest.mark.parametrize("name,tile,L", [("dirichlet_3level", 64, 4), ("neumann_3level", 48, 2), ("neumann_3level", 200, 4)])
def test_lds_resident_phase_kernel_and_12bit_slots_change_no_bit(name, tile, L):
    """Two layout/latency options of the sweep: (i) small levels run their phases with the tile's whole packed
    stream resident in LDS (tile_kernel_lds, default on), (ii) mmg_set_option("slot_bits", 12) packs the
    tile-local column indices in 12 bits.  Neither changes the order of any floating-point operation:
    iterates are bitwise those of the plain per-phase kernel with 16-bit slots, and follow the oracle."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    o.boundary_op(0)
    o.sor_sweeps(3)
    xs = {}
    try:
        for key, (res, bits) in {"plain": (0, 16), "lds": (1, 16), "lds12": (1, 12), "plain12": (0, 12)}.items():
            _capi.set_option("lds_resident", res)
            _capi.set_option("slot_bits", bits)
            _capi.set_option("persistent_sweep", 0)
            d = H.device_level(la, tile_size=tile, lanes_per_row=L)
            d.boundary_op(0)
            d.sweeps(3)
            xs[key] = d.get_x()
            assert H.rel_err(xs[key], o.x) < 1e-12, key
            assert abs(d.residual_ratio() - o.residual_ratio()) <= 1e-10 * o.residual_ratio(), key
    finally:
        _capi.set_option("lds_resident", 1)
        _capi.set_option("slot_bits", 12)
        _capi.set_option("persistent_sweep", 1)
    for key in ("lds", "lds12", "plain12"):
        assert np.array_equal(xs[key], xs["plain"]), key


@pytest.mark.parametrize("name", ["dirichlet_3level", "neumann_3level", "dirichlet_2level_inhomog"])
def test_resident_whole_sweep_kernel_matches_oracle_and_phase_launches(name):
    """Tiny levels (all tiles resident at once) run ALL phases and fused sweeps in one launch with the tile
    streams kept in LDS (sweep_resident_kernel; default for such levels).  Iterates: bitwise those of the
    per-phase launches, oracle to 1e-12; the V-cycle history (which now runs through this kernel on every
    level of the fixtures) to 1e-10."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    for lvl in range(case["nlevels"]):
        la = H.level_arrays(case, lvl)
        o = H.oracle_level(la)
        o.boundary_op(0)
        xs = {}
        try:
            for key, pers in (("phase", 0), ("resident", 1)):
                _capi.set_option("persistent_sweep", pers)
                d = H.device_level(la, tile_size=48, lanes_per_row=4)
                d.boundary_op(0)
                d.sweeps(7)                    # Dirichlet levels: one launch carrying 7 sweeps
                xs[key] = d.get_x()
        finally:
            _capi.set_option("persistent_sweep", 1)
        o.sor_sweeps(7)
        assert H.rel_err(xs["resident"], o.x) < 1e-12, lvl
        assert np.array_equal(xs["resident"], xs["phase"]), lvl
    om = H.oracle_multigrid(case)
    dh = H.device_hierarchy(case)
    for k in range(8):
        ro, rd = om.vcycle(), dh.vcycle()
        if ro < 0:
            continue
        assert abs(rd - ro) <= 1e-10 * ro + 2e-13, (k, rd, ro)


@pytest.mark.parametrize("name", ["dirichlet_3level", "neumann_3level", "dirichlet_2level_inhomog"])
def test_residual_rows_through_lds_change_no_bit(name):
    """Grid::residual (grid.cpp:147-151): by default the rows of a tile are collected in LDS and leave it in one
    coalesced pass; mmg_set_option("resid_lds", 0) stores them one by one.  Same r (bitwise, every entry incl.
    masked Dirichlet rows, Neumann rows and the multiplier row), same norms, and the oracle's r."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    for lvl in range(case["nlevels"]):
        la = H.level_arrays(case, lvl)
        la["x0"] = np.random.default_rng(lvl).standard_normal(la["a_size"])
        o = H.oracle_level(la)
        ro = o.residual()
        out = {}
        try:
            for opt in (0, 1):
                _capi.set_option("resid_lds", opt)
                d = H.device_level(la, tile_size=40, lanes_per_row=4)
                out[opt] = (d.residual(), d.residual_ratio())
        finally:
            _capi.set_option("resid_lds", 1)
        assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1], lvl
        assert H.rel_err(out[1][0], ro) < 1e-12, lvl


def test_lds_resident_phase_kernel_3d_k50():
    """The same on the 3-D K = 50 stencils of the coarse V-cycle levels (135 KB of stream per 256-point tile,
    more than 64 KiB of dynamic LDS per workgroup), including the single-launch sweep with 12-bit slots."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi, _host
    pts = _host.box_cloud(30, 3, seed=4)
    g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC, tile_points=256,
                                 lanes_per_row=2)
    la = g.level_arrays()
    la["b0"] = np.random.default_rng(3).standard_normal(la["a_size"])
    o = H.oracle_level(la)
    o.sor_sweeps(3)
    xs = {}
    try:
        for key, (res, bits, pers) in {"plain": (0, 16, 0), "lds": (1, 16, 0), "lds12": (1, 12, 0), "single12": (0, 12, 4)}.items():
            _capi.set_option("lds_resident", res)
            _capi.set_option("slot_bits", bits)
            _capi.set_option("persistent_sweep", pers)
            d = _capi.Level(la["n"], la["rowptr"], la["col"], la["val"], la["bcflags"], 0, 1.4, 5, la["btype"], la["bptr"],
                            la["bpts"], la["bvals"], x=la["x0"], b=la["b0"], tile_ptr=g.tile_ptr(), lanes_per_row=2)
            d.sweeps(3)
            xs[key] = d.get_x()
            assert H.rel_err(xs[key], o.x) < 1e-12, key
    finally:
        _capi.set_option("lds_resident", 1)
        _capi.set_option("slot_bits", 12)
        _capi.set_option("persistent_sweep", 1)
    for key in ("lds", "lds12", "single12"):
        assert np.array_equal(xs[key], xs["plain"]), key


@pytest.fixture
def exact_mode():
    from meshlessmultigridpoisson_amd import _capi
    _capi.set_option("exact_arithmetic", 1)
    yield
    _capi.set_option("exact_arithmetic", 0)


@pytest.mark.parametrize("name", CASES)
def test_exact_arithmetic_mode_is_bitwise_the_oracle(name, exact_mode):
    """mmg_set_option("exact_arithmetic", 1): one lane per row, stored order, separately
    rounded multiply/add.  Every iterate, residual vector and residual ratio must then be
    BITWISE the CPU oracle's -- for 20 V-cycles, far below the 1e-10 target at every cycle,
    including the stagnated tail where the fast kernels can only agree to the rounding floor."""
    _need_gpu()
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    d = H.device_level(la, tile_size=96)
    o.boundary_op(0)
    d.boundary_op(0)
    o.sor_sweeps(3)
    d.sweeps(3)
    assert np.array_equal(d.get_x(), o.x)
    assert np.array_equal(d.residual(), o.residual())
    assert d.residual_ratio() == o.residual_ratio()
    o.bound_eval_neumann()
    d.bound_eval_neumann()
    assert np.array_equal(d.get_x(), o.x)
    om = H.oracle_multigrid(case)
    dh = H.device_hierarchy(case)
    for k in range(len(case["resid_history"])):
        ro, rd = om.vcycle(), dh.vcycle()
        assert rd == ro, (k, rd, ro)
        assert rd == case["resid_history"][k]
    for lo, ld in zip(om.levels, dh.levels):
        assert np.array_equal(ld.get_x(), lo.x)
        assert np.array_equal(ld.get_rhs(), lo.b)


@pytest.mark.parametrize("name", ["neumann_2level", "dirichlet_3level"])
def test_distributed_code_path_single_rank(name):
    """Levels registered as distributed (mmg_level_set_exchange with an empty neighbour list,
    communicator of one rank): the V-cycle then runs the multi-GPU code path -- separate
    partial-sum / all-reduce / apply kernels for the multiplier row, the distributed residual
    finalisation, the halo hooks around restriction and prolongation -- and must still follow
    the oracle (with one part the hybrid schedule IS the sequential one)."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    case = H.load_case(name)
    om = H.oracle_multigrid(case)
    _capi.comm_init(0, 1, _capi.comm_unique_id())
    try:
        dh = H.device_hierarchy(case)
        for lv in dh.levels:
            lv.set_exchange(lv.n, [], [0], [], [0])
        for k in range(8):
            ro, rd = om.vcycle(), dh.vcycle()
            assert abs(rd - ro) <= 1e-10 * ro + max(FLOOR, H.rho_evaluation_noise(om.levels[-1])), (k, rd, ro)
        assert H.rel_err(dh.levels[-1].get_x(), om.levels[-1].x) < 1e-9
    finally:
        _capi.comm_finalize()


def test_edge_cases_empty_and_degenerate_levels():
    """Degenerate inputs: a level whose points are ALL Dirichlet (no row is ever relaxed), a
    level with a single interior point, a tile size far larger than the level."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi
    n = 9
    rowptr = np.arange(n + 1, dtype=np.int32)
    col = np.arange(n, dtype=np.int32)
    val = np.full(n, 2.0)
    flags = np.ones(n, dtype=np.int32)
    lv = _capi.Level(n, rowptr, col, val, flags, 0, 1.4, 5, [1], [0, n], np.arange(n), np.linspace(1, 2, n),
                     x=np.zeros(n), b=np.ones(n), tile_size=4096)
    lv.boundary_op(0)
    x0 = lv.get_x()
    lv.sor()
    assert np.array_equal(lv.get_x(), x0) and np.array_equal(x0, np.linspace(1, 2, n))
    assert not lv.residual().any()                       # every row masked (grid.cpp:149)
    assert lv.info()["sor_rows"] == 0
    # one interior point coupled to two Dirichlet points: x <- (1-w)x + w/a (b - sum)
    rowptr = np.array([0, 1, 4, 5], dtype=np.int32)
    col = np.array([0, 0, 1, 2, 2], dtype=np.int32)
    val = np.array([1.0, 1.0, -2.0, 1.0, 1.0])
    lv = _capi.Level(3, rowptr, col, val, [1, 0, 1], 0, 1.4, 5, [1], [0, 2], [0, 2], [3.0, 5.0],
                     x=np.zeros(3), b=np.array([0.0, 4.0, 0.0]))
    lv.boundary_op(0)
    lv.sweeps(1)
    want = (1 - 1.4) * 0.0 + 1.4 / -2.0 * (4.0 - (3.0 + 5.0))
    assert np.allclose(lv.get_x(), [3.0, want, 5.0], rtol=1e-15)
    assert np.isclose(lv.residual()[1], 4.0 - (3.0 - 2.0 * want + 5.0))
    with pytest.raises(_capi.MmgError):                  # sizes that do not fit together are rejected loudly
        _capi.Level(3, rowptr, col, val, [1, 0, 1], 1, 1.4, 5, [1], [0, 2], [0, 2], [3.0, 5.0])


def test_full_size_properties_1e7_points():
    """BASELINE configs[2] size (216^3 = 1.008e7 points, K = 50) on the operator bench.py times -- the reference's
    RBF-FD Laplacian (Grid::build_laplacian: PHS r^3 + degree-3 polynomials, stencil solves batched on the
    device) -- where the oracle is too slow to be the checker: size-independent properties of the relaxation
    and residual operators.
      * the single-launch (fused) sweep and the per-phase launches give identical bits;
      * one SOR sweep is LINEAR in (x, b):  S(x1 + x2, b1 + b2) = S(x1, b1) + S(x2, b2);
      * the residual is linear, and zero (to rounding) at a fixed point b := A x*, which the sweep
        then leaves unchanged (idempotence on exact solutions);
      * a spot check of 2000 random rows of the sweep against the row formula of grid.cpp:122-141
        evaluated in numpy from the CSR (Jacobi-style from the pre-sweep state is NOT what is
        checked: the rows are checked against the post-sweep neighbours they must have seen)."""
    _need_gpu()
    from meshlessmultigridpoisson_amd import _capi, _host
    pts = _host.box_cloud(216, 3, seed=12345)
    _host.set_option("device_setup", 1)
    try:
        g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_DIRICHLET, ordering=_host.ORDER_MC, tile_points=0)
    finally:
        _host.set_option("device_setup", -1)
    sz = g.sizes()
    n = sz["n"]
    lv = _capi.Level.borrow(g.device_level(), n, sz["a_size"])
    _xyz, flags = g.points()
    interior = flags == 0
    rng = np.random.default_rng(1)
    x1, x2 = rng.standard_normal(n) * interior, rng.standard_normal(n) * interior
    b1, b2 = rng.standard_normal(n), rng.standard_normal(n)

    def sweep(x, b, k=1, mode=1):
        _capi.set_option("persistent_sweep", mode)
        lv.set_x(x)
        lv.set_rhs(b)
        lv.sweeps(k)
        _capi.set_option("persistent_sweep", 1)
        return lv.get_x()

    s1, s2, s12 = sweep(x1, b1), sweep(x2, b2), sweep(x1 + x2, b1 + b2)
    scale = np.abs(s12).max()
    assert np.abs(s12 - (s1 + s2)).max() <= 1e-11 * scale
    assert np.array_equal(sweep(x1, b1, k=3, mode=1), sweep(x1, b1, k=3, mode=0))
    # residual: linear, and a fixed point stays fixed
    lv.set_x(x1)
    lv.set_rhs(b1)
    r1 = lv.residual()
    lv.set_x(x2)
    lv.set_rhs(b2)
    r2 = lv.residual()
    lv.set_x(x1 + x2)
    lv.set_rhs(b1 + b2)
    r12 = lv.residual()
    assert np.abs(r12 - (r1 + r2)).max() <= 1e-11 * np.abs(r12).max()
    lv.set_x(x1)
    lv.set_rhs(np.zeros(n))
    b_star = -lv.residual()                               # = A x1 on interior rows
    b_star[~interior] = 0.0
    x_fix = sweep(x1, b_star, k=2)
    assert np.abs(x_fix - x1).max() <= 1e-11 * np.abs(x1).max()
    lv.set_x(x1)
    lv.set_rhs(b_star)
    assert lv.residual_ratio() <= 1e-12
    # spot check rows against the reference's row formula with post-sweep neighbour values
    rowptr, col, val = g.csr()
    xs = sweep(x1, b1)
    rows = rng.choice(np.nonzero(interior)[0], size=2000, replace=False)
    for i in rows:
        c, v = col[rowptr[i]:rowptr[i + 1]], val[rowptr[i]:rowptr[i + 1]]
        d = v[c == i][0]
        off = c != i
        # a neighbour was already updated iff it is an interior row stored before i (Gauss-Seidel order)
        seen = np.where((c[off] < i) & interior[c[off]], xs[c[off]], x1[c[off]])
        want = (1 - 1.4) * x1[i] + 1.4 / d * (b1[i] - (v[off] * seen).sum())
        assert abs(xs[i] - want) <= 1e-11 * max(1.0, abs(want)), i


@pytest.mark.parametrize("n,max_len,dfrac", [(1, 1, 0.0), (2, 2, 0.0), (65, 3, 0.0), (300, 40, 0.2), (257, 200, 0.0), (120, 20, 1.0)])
@pytest.mark.parametrize("waves", [0, 1, 4])
def test_edge_case_levels_on_device(n, max_len, dfrac, waves):
    """SURVEY 8c edge cases through the C-ABI: a level of one point, of two, ragged rows of 1 ... 200 entries
    (diagonal-only rows included), 20 % Dirichlet points, a level of Dirichlet points only (nothing is relaxed, the
    masked residual is zero, the ratio is 0 / ||b|| = 0).  Automatic, packed and dense layouts; sweeps, residual vector
    and residual ratio equal the oracle's; zero sweeps change nothing."""
    _need_gpu()
    la = H.ragged_level(n, seed=n + max_len, max_len=max_len, dirichlet_frac=dfrac)
    o = H.oracle_level(la)
    d = H.device_level(la, tile_size=64, waves_per_tile=waves)
    o.boundary_op(0)
    d.boundary_op(0)
    assert np.array_equal(d.get_x(), o.x)
    d.sweeps(0)
    assert np.array_equal(d.get_x(), o.x)
    o.sor_sweeps(3)
    d.sweeps(3)
    assert H.rel_err(d.get_x(), o.x) < 1e-12
    ro, rd = o.residual(), d.residual()
    assert np.abs(rd - ro).max() <= 1e-11 * max(1.0, np.abs(ro).max())
    assert abs(d.residual_ratio() - o.residual_ratio()) <= 1e-10 * o.residual_ratio() + 1e-15
    info = d.info()
    assert info["sor_rows"] == int((la["bcflags"] == 0).sum())
