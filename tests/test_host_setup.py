"""Host C++ setup (csrc/host) against the oracle: leaf functions and file readers
against outputs of the REFERENCE's own sources (tests/golden/ref_utils.npz, made by
oracle/_ref), RBF-FD assembly against the numpy restatement, and the properties of
the MI355X ordering.  No GPU needed: nothing here touches the hot path."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import helpers as H

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


@pytest.fixture(scope="module")
def host():
    from meshlessmultigridpoisson_amd import _host
    return _host


@pytest.fixture(scope="module")
def refu():
    z = np.load(os.path.join(H.GOLDEN, "ref_utils.npz"))
    return {k: z[k] for k in z.files}


def test_distance_and_shifting_scaling_bitwise_vs_reference(host, refu):
    L = host.lib()
    for pq, want in zip(refu["dist_in"], refu["dist_out"]):
        p, q = np.ascontiguousarray(pq[0]), np.ascontiguousarray(pq[1])
        assert L.mmgh_distance(p.ctypes.data_as(_dp), q.ctypes.data_as(_dp)) == want
    pts = np.ascontiguousarray(refu["ss_pts"])
    ev = np.ascontiguousarray(refu["ss_eval"])
    out = np.zeros((len(pts) + 2, 3))
    L.mmgh_shifting_scaling(pts.ctypes.data_as(_dp), len(pts), ev.ctypes.data_as(_dp), out.ctypes.data_as(_dp))
    assert np.array_equal(out, refu["ss_out"])


def test_rcm_matches_reference(host, refu):
    ptr, idx = refu["rcm_ptr"], refu["rcm_idx"]
    order = np.zeros(len(ptr) - 1, dtype=np.int32)
    n = host.lib().mmgh_rcm(ptr.ctypes.data_as(_ip), idx.ctypes.data_as(_ip), len(ptr) - 1, order.ctypes.data_as(_ip))
    assert np.array_equal(order[:n], refu["rcm_order"])


def test_msh_reader_and_conn_match_reference(host, refu, tmp_path):
    msh = os.path.join(H.GOLDEN, "tiny_square.msh").encode()
    xyz = np.zeros((64, 3))
    n = host.lib().mmgh_points_from_msh(msh, xyz.ctypes.data_as(_dp), 64, 0)
    assert n == len(refu["msh_points"]) and np.array_equal(xyz[:n], refu["msh_points"])
    flags = np.ascontiguousarray(refu["msh_bcflags"])
    conn = np.zeros((n, 2), dtype=np.int32)
    host.lib().mmgh_bound_pts_conn(msh, flags.ctypes.data_as(_ip), n, conn.ctypes.data_as(_ip))
    assert np.array_equal(conn, refu["msh_conn"])
    # writer: same text as the reference's writeVectorToTxt (default ostream precision)
    v = np.ascontiguousarray(refu["txt_vec"])
    out = tmp_path / "v.txt"
    host.lib().mmgh_write_vector_txt(v.ctypes.data_as(_dp), len(v), str(out).encode())
    assert out.read_text() == open(os.path.join(H.GOLDEN, "ref_vector.txt")).read()
    # own MSH 2.2 writer round-trips through the reader; missing file -> empty, no crash
    pts = host.square_cloud(7, seed=3)
    f = str(tmp_path / "c.msh").encode()
    assert host.lib().mmgh_write_msh(f, pts.ctypes.data_as(_dp), len(pts)) == 0
    back = np.zeros((len(pts), 3))
    assert host.lib().mmgh_points_from_msh(f, back.ctypes.data_as(_dp), len(pts), 0) == len(pts)
    assert np.array_equal(back, pts)
    assert host.lib().mmgh_points_from_msh(b"/nonexistent.msh", back.ctypes.data_as(_dp), 1, 0) == 0
    assert host.lib().mmgh_order_from_txt(f, 3) == 0  # reference quirk: reads and returns nothing


def _oracle_grid(pts, polydeg, neumann):
    from oracle import setup_oracle as so
    props = so.make_props(polydeg)
    return (so.gen_grid_neumann_square(pts, props) if neumann else so.gen_grid_dirichlet_square(pts, props))


@pytest.mark.parametrize("neumann", [False, True])
def test_laplacian_assembly_matches_numpy_oracle(host, neumann):
    """Same cloud, same RCM ordering: identical sparsity, boundary lists and RHS;
    weights agree to 1e-7 of the row scale (two independent full-pivot LU codes on
    ill-conditioned PHS systems; the hot-path parity tests use ONE matrix for both
    sides, so this tolerance never enters them)."""
    pts = host.square_cloud(15, seed=11)
    og = _oracle_grid(pts, 3, neumann)
    hg = host.Grid.create_square(pts, 3, kind=host.KIND_NEUMANN if neumann else host.KIND_DIRICHLET,
                                 ordering=host.ORDER_RCM)
    rowptr, col, val = hg.csr()
    orp, ocol, oval = og.csr
    xyz, flags = hg.points()
    assert np.array_equal(xyz, og.points) and np.array_equal(flags, og.bcflags)
    assert np.array_equal(rowptr, orp) and np.array_equal(col, ocol)
    scale = np.abs(oval).max()
    assert np.abs(val - oval).max() <= 1e-7 * scale
    btype, bptr, bpts, bvals = hg.boundaries()
    obt, obp, obpts, obv = og.boundary_arrays()
    assert np.array_equal(bpts, obpts) and np.array_equal(btype, obt) and np.array_equal(bvals, obv)
    assert np.abs(hg.source() - og.source).max() <= 1e-6 * np.abs(og.source).max()


def test_transfer_matrices_match_numpy_oracle(host):
    from oracle import setup_oracle as so
    clouds = [host.square_cloud(9, seed=1), host.square_cloud(17, seed=2)]
    grids = [so.gen_grid_dirichlet_square(c, so.make_props(3)) for c in clouds]
    R, P = so.build_matrices(grids)
    mg = host.Multigrid(clouds, [3, 3], ordering=host.ORDER_RCM)
    for which, want in (("R", R[1]), ("P", P[0])):
        got = mg.transfer(which, 1 if which == "R" else 0)
        assert got["rows"] == want["rows"] and got["cols"] == want["cols"]
        assert np.array_equal(got["colptr"], want["colptr"]) and np.array_equal(got["rowidx"], want["rowidx"])
        assert np.abs(got["val"] - want["val"]).max() <= 1e-8


def test_knn_matches_bruteforce_with_reference_tiebreak(host):
    """Lattice clouds have many exact distance ties: order must be (distance, index)."""
    pts = host.square_cloud(12, seed=0, jitter=0.0)
    g = host.Grid.create_square(pts, 3, kind=host.KIND_GRAPH, ordering=host.ORDER_NONE)
    for pid in (0, 5, 77, 143):
        d = np.sqrt((pts[:, 0] - pts[pid, 0]) ** 2 + (pts[:, 1] - pts[pid, 1]) ** 2)
        want = np.lexsort((np.arange(len(pts)), d))[:25]
        assert np.array_equal(g.knn(pid, 25), want)


@pytest.mark.parametrize("dim,nside,poly,tile", [(2, 40, 3, 128), (3, 14, 2, 256)])
def test_mc_ordering_gives_few_phases_and_exact_schedule(host, dim, nside, poly, tile):
    """mc_order_points is a permutation; the plan built from the matrix in that order
    has few phases (tile colours) and, run by the adversarial CPU interpreter,
    reproduces the oracle's sequential sweep in the same order."""
    pts = host.box_cloud(nside, dim, seed=4)
    g = host.Grid.create_square(pts, poly, dim=dim, kind=host.KIND_GRAPH, ordering=host.ORDER_MC, tile_points=tile)
    xyz, _ = g.points()
    assert sorted(map(tuple, xyz.round(12))) == sorted(map(tuple, pts.round(12)))
    la = g.level_arrays()
    rng = np.random.default_rng(0)
    la["b0"] = rng.standard_normal(la["a_size"])
    e = H.EmuLevel(la, tile_ptr=g.tile_ptr(), lanes_per_row=4)
    info = e.info()
    assert info["n_tiles"] == g.sizes()["n_tiles"]
    assert info["n_phases"] <= (6 if dim == 2 else 14)
    o = H.oracle_level(la)
    o.sor_sweeps(2)
    e.sweeps(2)
    assert H.rel_err(e.x, o.x) < 1e-12
    # the same matrix in RCM order needs far more phases
    g2 = host.Grid.create_square(pts, poly, dim=dim, kind=host.KIND_GRAPH, ordering=host.ORDER_RCM)
    e2 = H.EmuLevel(g2.level_arrays(), tile_size=tile, lanes_per_row=4)
    assert e2.info()["n_phases"] > info["n_phases"]


def test_3d_rbf_weights_reproduce_polynomials(host):
    """3-D extension (no reference counterpart): Laplacian weights of polyDeg 2 must be
    exact on quadratics: L(x^2 + 2 y^2 - z^2 + xy) = 2 + 4 - 2 = 4 at interior points."""
    pts = host.box_cloud(9, 3, seed=8)
    g = host.Grid.create_square(pts, 2, dim=3, kind=host.KIND_DIRICHLET, ordering=host.ORDER_NONE)
    rowptr, col, val = g.csr()
    xyz, flags = g.points()
    u = xyz[:, 0] ** 2 + 2 * xyz[:, 1] ** 2 - xyz[:, 2] ** 2 + xyz[:, 0] * xyz[:, 1]
    import scipy.sparse as sp
    A = sp.csr_matrix((val, col, rowptr), shape=(len(u), len(u)))
    lap = A @ u
    assert np.abs(lap[flags == 0] - 4.0).max() < 1e-6


def test_fracstep_operators_match_numpy_oracle_and_kovasznay(host):
    """FractionalStepGrid setup (fractionalStepGrid.cpp:60-100): D_x, D_y, Laplacian rows for
    every point match the numpy restatement; applied to the Kovasznay field they approximate
    the analytic derivatives (the reference's check_derivs, FractionalStepSim.cpp:80-103)."""
    from oracle import setup_oracle as so
    pts = host.square_cloud(21, seed=4)
    g = host.FracStepGrid.create(pts, polydeg=3, ordering=host.ORDER_RCM)
    xyz, flags = g.points()
    og = so.gen_grid_neumann_square(pts, so.make_props(3))   # same cloud, same RCM order
    assert np.array_equal(xyz, og.points)
    want = og.build_fs_matrices()
    for which in range(3):
        rp, col, val = g.op(which)
        assert np.array_equal(rp, want[which][0]) and np.array_equal(col, want[which][1])
        assert np.abs(val - want[which][2]).max() <= 1e-7 * np.abs(want[which][2]).max()
    import scipy.sparse as sp
    re = 1.0 / 0.025
    lam = 0.5 * re - np.sqrt(0.25 * re * re + 4 * np.pi ** 2)
    u = 1 - np.exp(lam * xyz[:, 0]) * np.cos(2 * np.pi * xyz[:, 1])
    dx = sp.csr_matrix((g.op(0)[2], g.op(0)[1], g.op(0)[0]), shape=(len(u), len(u)))
    exact = -lam * np.exp(lam * xyz[:, 0]) * np.cos(2 * np.pi * xyz[:, 1])
    assert np.abs(dx @ u - exact)[flags == 0].mean() < 5e-2


def test_binary_cloud_container_round_trip(tmp_path):
    """binary cloud format for 1e7+ points next to the reference's MSH / txt readers.
    Bitwise round trip, equal to what the MSH reader returns for the same cloud, loud on bad input."""
    from meshlessmultigridpoisson_amd import _host as host
    pts = host.box_cloud(9, 3, seed=3)
    f = str(tmp_path / "cloud.mmgc")
    host.write_cloud_bin(f, pts, 3)
    assert os.path.getsize(f) == 24 + 24 * len(pts)
    back, dim = host.read_cloud_bin(f)
    assert dim == 3 and np.array_equal(back, pts)
    # the same cloud through the reference's text format: %.17g text -> identical doubles
    msh = str(tmp_path / "cloud.msh").encode()
    assert host.lib().mmgh_write_msh(msh, pts.ctypes.data_as(_dp), len(pts)) == 0
    txt = np.zeros_like(pts)
    assert host.lib().mmgh_points_from_msh(msh, txt.ctypes.data_as(_dp), len(pts), 0) == len(pts)
    assert np.array_equal(txt, back)
    # truncated, foreign and missing files are refused
    raw = open(f, "rb").read()
    open(f, "wb").write(raw[:-8])
    with pytest.raises(host.HostError):
        host.read_cloud_bin(f)
    open(f, "wb").write(b"NOTCLOUD" + raw[8:])
    with pytest.raises(host.HostError):
        host.read_cloud_bin(f)
    with pytest.raises(host.HostError):
        host.read_cloud_bin(str(tmp_path / "missing.mmgc"))
    # a 2-D cloud through a Grid: same operator as from the in-memory points
    p2 = host.square_cloud(17, seed=2)
    host.write_cloud_bin(f, p2, 2)
    q2, d2 = host.read_cloud_bin(f)
    assert d2 == 2
    ga = host.Grid.create_square(p2, 3, ordering=host.ORDER_NONE)
    gb = host.Grid.create_square(q2, 3, ordering=host.ORDER_NONE)
    assert all(np.array_equal(a, b) for a, b in zip(ga.csr(), gb.csr()))


def test_neumann_3d_scaled_multiplier_row_and_edge_free_cloud_converge():
    """3-D Neumann hierarchies (no reference counterpart).  Two things the 2-D formulation needs in 3-D:
    the multiplier ROW carries n^(-1/3) instead of 1 (relaxing the bordered system couples the mean of x and the
    multiplier with strength n/|a_ii| ~ points per side: the constant mode explodes otherwise), and the box cloud has
    no nodes on its edges and corners (their one-sided n.grad rows are nearly singular).  With both, the CPU oracle's
    V-cycle contracts on a 10^3 / 19^3 hierarchy; the packed-plan interpreter (the device's arithmetic) follows the
    oracle through sweeps and residuals with the scaled row."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    clouds = [host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate([10, 19])]
    assert len(clouds[1]) == 19 ** 3 - 12 * 17 - 8                 # 12 edges of 17 inner nodes, 8 corners
    mg = host.Multigrid(clouds, [2, 2], dim=3, neumann=True, ordering=host.ORDER_MC, tile_points=96)
    la = mg.grid(1).level_arrays()
    n, rp, col, val = la["n"], la["rowptr"], la["col"], la["val"]
    row = slice(rp[n], rp[n + 1])
    offd = val[row][col[row] != n]
    assert np.all(offd == 1.0 / np.cbrt(float(n))) and val[row][col[row] == n][0] == 1.0
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    assert np.all(val[(col == n) & (rows != n)] == 1.0)            # the column keeps the reference's ones
    om = H.oracle_of_multigrid(mg)
    hist = [om.vcycle() for _ in range(14)]
    assert hist[-1] < 0.25 * hist[0] and hist[-1] < hist[-4], hist
    lv = H.oracle_level(la)
    emu = H.EmuLevel(la, tile_ptr=mg.grid(1).tile_ptr(), lanes_per_row=4)
    rng = np.random.default_rng(3)
    x0 = rng.standard_normal(len(lv.x))
    lv.x[:] = x0
    emu.x[:] = x0
    lv.sor_sweeps(3)
    emu.sweeps(3)
    assert H.rel_err(emu.x, lv.x) < 1e-12
    r_o, _ = lv.residual(), None
    r_e, _nrm = emu.residual()
    assert H.rel_err(r_e, r_o) < 1e-11


def test_annulus_known_answer_two_dirichlet_boundaries():
    """The reference's "concentric_circles" problem (testing_functions.cpp:107-135, calc_l1_error_circle :34-67): annulus
    0.25 <= r <= 0.5, homogeneous Dirichlet data on BOTH circles (two Boundary objects), manufactured solution
    sin(pi k r*).  Host-built three-level hierarchy, CPU oracle V-cycles: contracts, L1 error per point at the
    discretisation level (the reference prints it; it holds no number)."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    clouds = [host.annulus_cloud(nr, seed=12345 + i) for i, nr in enumerate([6, 12, 24])]
    r2 = (clouds[-1][:, 0] - 0.5) ** 2 + (clouds[-1][:, 1] - 0.5) ** 2
    assert (np.abs(0.25 - r2) <= 1e-10).sum() > 100 and (np.abs(0.0625 - r2) <= 1e-10).sum() > 50
    mg = host.Multigrid.annulus(clouds, [3, 3, 3], k=1, tile_points=128)
    g = mg.grid(2)
    assert g.sizes()["nb"] == 2
    om = H.oracle_of_multigrid(mg)
    hist = [om.vcycle() for _ in range(60)]
    assert hist[-1] < 1e-4 * hist[0]
    xyz, _fl = g.points()
    n = g.sizes()["n"]
    rstar = (np.sqrt((xyz[:, 0] - 0.5) ** 2 + (xyz[:, 1] - 0.5) ** 2) - 0.25) / 0.25
    err = np.abs(om.levels[-1].x[:n] - np.sin(np.pi * rstar)).sum() / n
    assert err < 5e-3, err                       # measured 1.3e-3 (24 rings, polyDeg 3)


def test_square_with_circle_known_answer_inhomogeneous_inner_boundary():
    """The reference's "square_with_circle" problem (testing_functions.cpp:85-106): unit square with a hole of radius
    0.25, u = 0 on the square and u = sin(k pi x) sin(k pi y) ON THE CIRCLE (a second boundary with non-zero values:
    boundaryOp "fine" writes them, "coarse" zeroes them on the coarse levels).  CPU oracle on the host-built hierarchy."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    clouds = [host.square_with_circle_cloud(n, seed=12345 + i) for i, n in enumerate([17, 33, 65])]
    mg = host.Multigrid.square_with_circle(clouds, [3, 3, 3], k=1, tile_points=128)
    g = mg.grid(2)
    assert g.sizes()["nb"] == 2
    om = H.oracle_of_multigrid(mg)
    hist = [om.vcycle() for _ in range(50)]
    assert hist[-1] < 1e-7 * hist[0]
    xyz, _fl = g.points()
    n = g.sizes()["n"]
    exact = np.sin(np.pi * xyz[:, 0]) * np.sin(np.pi * xyz[:, 1])
    on_circle = np.abs(0.0625 - (xyz[:, 0] - 0.5) ** 2 - (xyz[:, 1] - 0.5) ** 2) <= 1e-10
    assert on_circle.sum() > 50 and np.allclose(om.levels[-1].x[:n][on_circle], exact[on_circle], rtol=0, atol=1e-15)
    assert np.abs(om.levels[-1].x[:n] - exact).sum() / n < 1e-4       # measured 1.0e-5


def test_annulus_neumann_known_answer_inhomogeneous_data_on_curved_boundaries():
    """The reference's Neumann run on "concentric_circles" (testing_functions.cpp:212-250, the geometry its run_tests()
    loops over): radial inward normals (build_normal_vecs), NON-ZERO normal-derivative data on both circles
    (push_inhomog_to_rhs, modify_coeff_neumann), multiplier row.  Two-level hierarchy, CPU oracle: the residual drops
    three orders at once and then creeps (0.98 per cycle, the coarse grid is only smoothed); after the mean shift of
    calc_l1_error_circle the solution is sin(pi k r*) to the discretisation error."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    clouds = [host.annulus_cloud(nr, seed=12345 + i) for i, nr in enumerate([12, 24])]
    mg = host.Multigrid.annulus_neumann(clouds, [3, 3], k=1, tile_points=128)
    g = mg.grid(1)
    assert g.sizes()["nb"] == 2 and g.sizes()["neumann"] == 1
    om = H.oracle_of_multigrid(mg)
    hist = [om.vcycle() for _ in range(200)]
    assert hist[-1] < 2e-4 * hist[0]
    xyz, _fl = g.points()
    n = g.sizes()["n"]
    rstar = (np.sqrt((xyz[:, 0] - 0.5) ** 2 + (xyz[:, 1] - 0.5) ** 2) - 0.25) / 0.25
    exact = np.sin(np.pi * rstar)
    v = om.levels[-1].x[:n]
    err = np.abs(v + (exact.mean() - v.mean()) - exact).sum() / n
    assert err < 2e-3, err                       # measured 6.4e-4 after 200 cycles (3.0e-3 after 120: the slow tail)


def test_square_with_circle_neumann_known_answer():
    """The reference's Neumann run on "square_with_circle" (testing_functions.cpp:186-209): face normals on the square,
    radial normals on the circle, non-zero data on the circle only.  Two levels, CPU oracle; cos cos after the mean shift."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    clouds = [host.square_with_circle_cloud(n, seed=12345 + i) for i, n in enumerate([33, 65])]
    mg = host.Multigrid.square_with_circle_neumann(clouds, [3, 3], k=1, tile_points=128)
    g = mg.grid(1)
    om = H.oracle_of_multigrid(mg)
    hist = [om.vcycle() for _ in range(200)]
    assert hist[-1] < 2e-4 * hist[0]
    xyz, _fl = g.points()
    n = g.sizes()["n"]
    exact = np.cos(np.pi * xyz[:, 0]) * np.cos(np.pi * xyz[:, 1])
    v = om.levels[-1].x[:n]
    assert np.abs(v + (exact.mean() - v.mean()) - exact).sum() / n < 5e-4       # measured 6.8e-5


def test_damped_coarse_correction_makes_multilevel_neumann_contract():
    """Opt-in safeguard, NOT in the reference: x_f += theta * P x_c.  With theta = 1 (multigrid.cpp:102-106) the
    four-level Neumann cycle on 13^2 ... 97^2 diverges (x 30 per cycle); with theta = 0.7 it contracts.  CPU oracle
    (orc_vcycle_damped; theta = 1 is bitwise orc_vcycle)."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    host.set_option("point_colouring", 1)     # colour classes inside the tiles (round 2's order); the jittered cloud and
    try:                                       # this order together give the divergent cycle the safeguard was built for
        clouds = [host.square_cloud(n, seed=777 + i) for i, n in enumerate([13, 25, 49, 97])]
        mg = host.Multigrid(clouds, [3] * 4, neumann=True, ordering=host.ORDER_MC, tile_points=128)
        mg2 = host.Multigrid(clouds, [3] * 4, neumann=True, ordering=host.ORDER_MC, tile_points=128)
    finally:
        host.set_option("point_colouring", -1)
    plain = H.oracle_of_multigrid(mg)
    hist = [plain.vcycle() for _ in range(12)]
    assert hist[-1] > 1e3 * hist[0]                              # divergent here (x 30 per 5 cycles)
    mg2.damping = 0.7
    damped = H.oracle_of_multigrid(mg2)
    hist = [damped.vcycle() for _ in range(40)]
    assert hist[-1] < 2e-2 * hist[0] and hist[-1] < hist[-5], hist[-6:]
