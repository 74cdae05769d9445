"""The CPU oracle against the committed golden vectors (tests/golden/*.npz) and the
reference's known-answer content.  No GPU."""
import numpy as np
import pytest

from tests import helpers as H

CASES = ["dirichlet_3level", "neumann_2level", "neumann_3level", "dirichlet_2level_inhomog"]


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
