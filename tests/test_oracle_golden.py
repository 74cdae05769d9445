"""The CPU oracle against the committed golden vectors (tests/golden/*.npz) and the
reference's known-answer content.  No GPU."""
import numpy as np
import pytest

from tests import helpers as H

CASES = ["dirichlet_3level", "neumann_2level", "neumann_3level", "dirichlet_2level_inhomog", "neumann_live_L6_3level"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden_history(name):
    case = H.load_case(name)
    mg = H.oracle_multigrid(case)
    for _ in range(len(case["resid_history"])):
        mg.vcycle()
    assert np.array_equal(np.array(mg.residuals), case["resid_history"])  # same code, same machine class: bitwise
    assert np.array_equal(mg.levels[-1].x, case[f"L{case['nlevels'] - 1}_x_final"])


@pytest.mark.parametrize("name", CASES)
def test_oracle_single_level_goldens(name):
    case = H.load_case(name)
    lv = H.oracle_level(H.level_arrays(case, case["nlevels"] - 1))
    lv.boundary_op(0)
    assert np.array_equal(lv.residual(), case["fine_resid0"])
    lv.sor_sweeps(1)
    assert np.array_equal(lv.x, case["fine_x_after_1sweep"])
    lv.sor_sweeps(lv.iters - 1)
    assert np.array_equal(lv.x, case["fine_x_after_sor"])


def test_residuals_bookkeeping_is_before_cycle_N1():
    """multigrid.cpp:66-67: residuals_[k] is the relative L1 residual BEFORE cycle k."""
    case = H.load_case("dirichlet_3level")
    mg = H.oracle_multigrid(case)
    r0 = mg.residual()
    assert mg.vcycle() == r0
    r1 = mg.residual()
    assert mg.vcycle() == r1


def test_known_answer_manufactured_solution():
    """testing_functions.cpp:85 / :3-33: Dirichlet sin(pi x) sin(pi y); the discrete
    solution converges to the manufactured one at discretisation level."""
    case = H.load_case("dirichlet_3level")
    mg = H.oracle_multigrid(case)
    for _ in range(30):
        mg.vcycle()
    pts = case["L2_points"]
    exact = np.sin(np.pi * pts[:, 0]) * np.sin(np.pi * pts[:, 1])
    assert np.abs(mg.levels[-1].x - exact).sum() / len(exact) < 2e-5
    assert mg.residuals[-1] < 1e-9


def test_quirk_N6_two_level_inhomogeneous_dirichlet():
    """multigrid.cpp:91 zeroes the FINE grid's Dirichlet values before post-smoothing when
    there are two levels; they are restored by boundaryOp("fine") at the next cycle."""
    case = H.load_case("dirichlet_2level_inhomog")
    mg = H.oracle_multigrid(case)
    mg.vcycle()
    fine = mg.levels[1]
    assert np.all(fine.x[fine.bpts] == 0.0)
    assert np.any(fine.bvals != 0.0)


def test_fracstep_single_grid_early_out():
    """FracStepMultigrid.cpp:64-67: one grid -> vCycle is one sor(), nothing pushed."""
    from oracle import oracle_c as oc
    case = H.load_case("neumann_2level")
    la = H.level_arrays(case, 1)
    a, b = H.oracle_level(la), H.oracle_level(la)
    mg = oc.Multigrid([a], [None], [None], frac_step=True)
    assert mg.vcycle() == -1.0 and mg.residuals == []
    b.sor()
    assert np.array_equal(a.x, b.x)


def test_hybrid_schedule_equals_sequential_for_one_part():
    case = H.load_case("neumann_2level")
    la = H.level_arrays(case, 1)
    a, b = H.oracle_level(la), H.oracle_level(la)
    a.sor_sweeps(3)
    b.sor_hybrid(np.zeros(la["n"], dtype=np.int32), 1, 3)
    assert np.array_equal(a.x, b.x)
    c = H.oracle_level(la)
    c.sor_hybrid((np.arange(la["n"]) * 2 // la["n"]).astype(np.int32), 2, 3)
    n = la["n"]
    assert not np.array_equal(a.x, c.x) and np.abs(a.x[:n] - c.x[:n]).max() < 0.5 * np.abs(a.x[:n]).max()


def test_oracle_3d_fracstep_statements_against_numpy():
    """The 3-D extension of the oracle (orc_fs_*3) and orc_push_inhomog (grid.cpp:664-685), checked against the
    same statements written with scipy / numpy on random operators -- pins the C restatement independently of
    the product."""
    import scipy.sparse as sp
    from oracle import oracle_c as oc
    rng = np.random.default_rng(4)
    n = 60
    mats = [sp.random(n, n, density=0.2, random_state=int(s), format="csr") for s in (1, 2, 3, 4)]
    csr = [(m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.copy()) for m in mats]
    nx, ny, nz = rng.standard_normal((3, n))
    bpts = np.arange(0, n, 7, dtype=np.int32)
    o = oc.FracStep3(n, csr[0], csr[1], csr[2], csr[3], nx, ny, nz, bpts)
    o.u[:], o.v[:], o.w[:] = rng.standard_normal((3, n))
    dt, mu, rho = 1e-2, 0.3, 1.7
    o.calc_hat(dt, mu, rho)
    dx, dy, dz, lap = mats
    for c, h in ((o.u, o.u_hat), (o.v, o.v_hat), (o.w, o.w_hat)):
        want = c + dt * (-(o.u * (dx @ c) + o.v * (dy @ c) + o.w * (dz @ c)) + mu / rho * (lap @ c))
        assert np.allclose(h, want, rtol=1e-13, atol=1e-14)
    src = np.zeros(n + 1)
    o.set_ppe_source(src, dt, rho)
    want = rho / dt * (dx @ o.u_hat + dy @ o.v_hat + dz @ o.w_hat)
    g = -rho / dt * np.stack([o.u - o.u_hat, o.v - o.v_hat, o.w - o.w_hat])
    want[bpts] = (nx * g[0] + ny * g[1] + nz * g[2])[bpts]
    assert np.allclose(src[:n], want, rtol=1e-12, atol=1e-12)
    p = rng.standard_normal(n)
    o.correct(p, dt, rho)
    assert np.allclose(o.w, o.w_hat - dt / rho * (dz @ p), rtol=1e-13, atol=1e-14)
    # push_inhomog: b_i -= sum_j A_ij * b_j / a_jj over interior rows i
    flags = rng.integers(0, 3, n).astype(np.int32)
    diag = 1.0 + rng.random(n)
    b = rng.standard_normal(n + 1)
    want = b.copy()
    C = mats[0]
    for i in range(n):
        if flags[i] == 0:
            for q in range(C.indptr[i], C.indptr[i + 1]):
                want[i] -= C.data[q] * b[C.indices[q]] / diag[C.indices[q]]
    oc.push_inhomog(n, csr[0], diag, flags, b)
    assert np.allclose(b, want, rtol=1e-14, atol=1e-15)


def test_colour_parallel_cpu_sweep_is_bitwise_the_sequential_one():
    """bench.py's optional all-cores CPU figure ("baseline only"): the tiles of one colour of the port's multicolour
    ordering relaxed concurrently on POSIX threads must give exactly the reference's sequential sweep."""
    from meshlessmultigridpoisson_amd import _host as host
    pts = host.box_cloud(14, 3, seed=3)
    g = host.Grid.create_square(pts, 2, dim=3, kind=host.KIND_GRAPH, ordering=host.ORDER_MC, tile_points=128)
    la = g.level_arrays()
    la["b0"] = np.random.default_rng(1).standard_normal(la["a_size"])
    tp = g.tile_ptr()
    ph, _gm = H.EmuLevel(la, tile_ptr=tp, lanes_per_row=4).point_phases()
    tile_phase = np.array([max(0, ph[tp[t]:tp[t + 1]].max(initial=0)) for t in range(len(tp) - 1)], dtype=np.int32)
    seq, par = H.oracle_level(la), H.oracle_level(la)
    seq.sor_sweeps(3)
    par.sor_sweeps_tiled(3, tp, tile_phase, 4)
    assert np.array_equal(seq.x, par.x)

