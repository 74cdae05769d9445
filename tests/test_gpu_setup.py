"""GPU tests of the setup kernels: mmg_knn replaces Grid::kNearestNeighbors (grid.cpp:216-260,
identical lists bit for bit), mmg_rbf_weights / mmg_rbf_stencils the per-point fullPivLu().solve of
Grid::laplaceWeights / derivx_weights / derivy_weights / pointInterpWeights (grid.cpp:263-424, :687-712).

Parity status: the reference ships no fixture for its setup; the checker here is the build's own
host restatement of those functions (csrc/host/grid.cpp, itself checked against the numpy
restatement oracle/setup_oracle.py in test_host_setup.py) plus the defining property of the
stencils -- exact reproduction of every polynomial of degree <= polyDeg.

Tolerance: both sides factorise the same saddle systems with full pivoting in fp64; they differ in
the rounding of pow()/products while the system is assembled, amplified by the conditioning of the
scaled saddle system (1e6..1e9 for these stencils).  Weights therefore agree to 1e-6 of the row's
largest weight; the polynomial reproduction, which does not see the conditioning, holds to 1e-8."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host():
    from meshlessmultigridpoisson_amd import _capi, _host
    assert _capi.device_count() >= 1, "no HIP device visible: libmmgp has no CPU fallback"
    yield _host
    _host.set_option("device_setup", -1)


def _monomials(dim, deg):
    out = []
    for p in range(deg + 1):
        for q in range(p + 1):
            if dim < 3:
                out.append((p - q, q, 0))
            else:
                for s in range(q + 1):
                    out.append((p - q, q - s, s))
    return out


def _csr_rows(g):
    rp, col, val = g.csr()
    return rp, col, val


@pytest.mark.parametrize("dim,nside,deg", [(2, 40, 3), (2, 40, 4), (2, 30, 5), (2, 26, 6), (3, 12, 3)])  # deg 6: 98 x 98 system, 77 KB of LDS per workgroup
def test_device_laplacian_matches_host_setup(host, dim, nside, deg):
    pts = host.box_cloud(nside, dim, seed=12345) if dim == 3 else host.square_cloud(nside, seed=12345)
    host.set_option("device_setup", 0)
    gh = host.Grid.create_square(pts, deg, dim=dim, ordering=host.ORDER_MC, tile_points=128)
    host.set_option("device_setup", 1)
    gd = host.Grid.create_square(pts, deg, dim=dim, ordering=host.ORDER_MC, tile_points=128)
    host.set_option("device_setup", -1)
    rph, colh, valh = gh.csr()
    rpd, cold, vald = gd.csr()
    assert np.array_equal(rph, rpd) and np.array_equal(colh, cold)      # same kNN pattern, same storage order
    n = len(rph) - 1
    rowmax = np.maximum.reduceat(np.abs(valh), rph[:-1])
    err = np.abs(vald - valh) / np.repeat(rowmax, np.diff(rph))
    assert err.max() <= 1e-6, err.max()
    # defining property: sum_j w_j p(x_j) = (laplace p)(x_i) for every monomial p up to the degree
    xyz, flags = gd.points()
    for (a, b, c) in _monomials(dim, deg):
        p = xyz[:, 0] ** a * xyz[:, 1] ** b * (xyz[:, 2] ** c if dim == 3 else 1.0)
        lap = np.zeros(n)
        if a >= 2:
            lap += a * (a - 1) * xyz[:, 0] ** (a - 2) * xyz[:, 1] ** b * (xyz[:, 2] ** c if dim == 3 else 1.0)
        if b >= 2:
            lap += b * (b - 1) * xyz[:, 0] ** a * xyz[:, 1] ** (b - 2) * (xyz[:, 2] ** c if dim == 3 else 1.0)
        if dim == 3 and c >= 2:
            lap += c * (c - 1) * xyz[:, 0] ** a * xyz[:, 1] ** b * xyz[:, 2] ** (c - 2)
        got = np.add.reduceat(vald * p[cold], rpd[:-1])
        scale = np.add.reduceat(np.abs(vald * p[cold]), rpd[:-1]) + 1.0
        assert (np.abs(got - lap) / scale).max() <= 1e-8, (a, b, c)


def test_device_rbf_weights_all_operators(host):
    """Every operator id against the host stencil functions, several right-hand sides on one factorisation."""
    from meshlessmultigridpoisson_amd import _capi
    pts = host.square_cloud(30, seed=7)
    host.set_option("device_setup", 0)
    fs = host.FracStepGrid.create(pts, polydeg=3, ordering=host.ORDER_NONE)
    host.set_option("device_setup", -1)
    xyz, flags = fs.points()
    n = len(xyz)
    ss = host.stencil_size(3, 2)
    interior = np.flatnonzero(flags == 0)
    nbr = np.stack([fs.knn(int(i), ss) for i in interior]).astype(np.int32)
    w = _capi.rbf_weights(2, 3, 3.0, xyz, xyz[interior], nbr, [1, 2, 0])   # d/dx, d/dy, laplace
    for o, which in enumerate((0, 1, 2)):                                   # FracStepGrid ops: D_x, D_y, velocity laplacian
        rp, col, val = fs.op(which)
        for k, i in enumerate(interior[:200]):
            row = dict(zip(col[rp[i]:rp[i + 1]], val[rp[i]:rp[i + 1]]))
            ref = np.array([row.get(int(c), 0.0) for c in nbr[k]])
            assert np.abs(w[o, k] - ref).max() <= 1e-6 * np.abs(ref).max(), (which, i)


def test_device_setup_fracstep_operators_match_host_setup(host):
    """FractionalStepGrid::build_derivX/derivY/uvLaplace (fractionalStepGrid.cpp:60-100) through the device
    batch: same sparsity, weights to 1e-6 of the row's largest entry, and one fractional step
    (predictor, PPE source, corrector) gives the same fields."""
    pts = host.square_cloud(28, seed=5)
    grids = []
    for mode in (0, 1):
        host.set_option("device_setup", mode)
        grids.append(host.FracStepGrid.create(pts, polydeg=3, ordering=host.ORDER_MC, tile_points=128))
    host.set_option("device_setup", -1)
    gh, gd = grids
    for which in (0, 1, 2):
        rph, colh, valh = gh.op(which)
        rpd, cold, vald = gd.op(which)
        assert np.array_equal(rph, rpd) and np.array_equal(colh, cold)
        rowmax = np.maximum.reduceat(np.abs(valh), rph[:-1])
        assert (np.abs(vald - valh) / np.repeat(rowmax, np.diff(rph))).max() <= 1e-6, which
    for g in (gh, gd):
        g.prescribe_soln()
        g.set_uv_bound()
        g.calc_hat()
        g.set_ppe_source()
    for which in (2, 3):          # u_hat, v_hat
        a, b = gh.vec(which), gd.vec(which)
        assert np.abs(a - b).max() <= 1e-7 * max(np.abs(a).max(), 1.0)
    assert np.abs(gh.source() - gd.source()).max() <= 1e-6 * np.abs(gh.source()).max()


@pytest.mark.parametrize("sides", [[13, 25, 49], [25, 49, 97]])   # 97^2 rows: the threaded column-major assembly
def test_device_setup_multigrid_converges_like_host_setup(host, sides):
    """Interpolation matrices + level operators from the device batch (neighbour search, solves, column-major
    assembly by counting transposition): identical sparsity to the host-setup hierarchy (host search, triplets),
    values equal to ~1e-7; the V-cycle history follows it and reaches the manufactured solution."""
    clouds = [host.square_cloud(n, seed=12345 + i) for i, n in enumerate(sides)]
    host.set_option("device_setup", 0)
    mh = host.Multigrid(clouds, [3, 3, 4], tile_points=128)
    host.set_option("device_setup", 1)
    md = host.Multigrid(clouds, [3, 3, 4], tile_points=128)
    host.set_option("device_setup", -1)
    for which in ("R", "P"):
        for l in range(3):
            th, td = mh.transfer(which, l), md.transfer(which, l)
            if th is None:
                assert td is None
                continue
            assert np.array_equal(th["colptr"], td["colptr"]) and np.array_equal(th["rowidx"], td["rowidx"])
            assert np.abs(th["val"] - td["val"]).max() <= 1e-7 * np.abs(th["val"]).max()
    rh = [mh.vcycle() for _ in range(10)]
    rd = [md.vcycle() for _ in range(10)]
    assert np.allclose(rd, rh, rtol=1e-4, atol=1e-12)
    # (the reference's V-cycle contracts more slowly on larger clouds: 0.53 after ten cycles at 97 x 97)
    assert rd[-1] < (0.05 if sides[-1] <= 49 else 0.6) * rd[0]


def test_rbf_weights_rejects_bad_input(host):
    from meshlessmultigridpoisson_amd import _capi
    xyz = host.square_cloud(10, seed=1)
    nbr = np.zeros((4, host.stencil_size(3, 2)), dtype=np.int32)
    nbr[2, 3] = len(xyz)   # out of range: must be caught on the host, never dereferenced on the device
    with pytest.raises(_capi.MmgError):
        _capi.rbf_weights(2, 3, 3.0, xyz, xyz[:4], nbr, [0])
    nbr[2, 3] = 0
    with pytest.raises(_capi.MmgError):
        _capi.rbf_weights(2, 3, 3.0, xyz, xyz[:4], nbr, [3])   # d/dz in 2-D


def test_device_rbf_weights_directly_against_numpy_oracle(host):
    """SURVEY 8f-2, without the product's host code in between: the stencil weights of the HIP kernel
    (mmg_rbf_weights) against oracle/setup_oracle.py -- the numpy restatement of kNearestNeighbors
    (grid.cpp:213-260), buildCoeffMatrix (:263-303), laplaceWeights (:381-424), derivx/derivy_weights
    (:304-380) and pointInterpWeights (:687-712) -- on the oracle's OWN neighbour lists, for interior and boundary
    evaluation points of a jittered 21 x 21 cloud, polyDeg 3 and 4.  The product's kNN (host cell grid) must
    return the oracle's lists.  Tolerance 1e-6 of the row's largest weight (conditioning of the saddle systems)."""
    from meshlessmultigridpoisson_amd import _capi
    from oracle import setup_oracle as so
    pts = host.square_cloud(21, seed=9)
    rng = np.random.default_rng(0)
    for deg in (3, 4):
        bp = [i for i, (x, y, _z) in enumerate(pts) if x in (0.0, 1.0) or y in (0.0, 1.0)]
        og = so.Grid(pts, [so.Boundary(1, bp, [0.0] * len(bp))], so.make_props(deg), np.zeros(len(pts)))
        og.set_bc_flag(0, "dirichlet", [0.0] * len(bp))     # the factory of testing_functions.cpp:68-159 up to the ordering
        ss = so.stencil_size(deg)
        ids = np.concatenate([rng.choice(np.flatnonzero(og.bcflags == 0), 40, replace=False),
                              np.flatnonzero(og.bcflags != 0)[:10]])
        nbr = np.array([og.k_nearest(og.points[i], og.neumann, og.bcflags[i] != 0, ss) for i in ids], dtype=np.int32)
        # the product's kNN returns the same lists (distance, index order)
        g = host.Grid.create_square(og.points, deg, kind=host.KIND_GRAPH, ordering=host.ORDER_NONE)
        for k, i in enumerate(ids):
            if og.bcflags[i] == 0:
                assert np.array_equal(g.knn(int(i), ss), nbr[k]), i
        w = _capi.rbf_weights(2, deg, 3.0, og.points, og.points[ids], nbr, [0, 1, 2])   # laplace, d/dx, d/dy
        for k, i in enumerate(ids):
            refs = [og.laplace_weights(int(i))[0][:ss], og.deriv_weights(int(i), 0)[0][:ss], og.deriv_weights(int(i), 1)[0][:ss]]
            for o, ref in enumerate(refs):
                assert np.abs(w[o, k] - ref).max() <= 1e-6 * np.abs(ref).max(), (deg, int(i), o)
        # interpolation weights at off-cloud points (Multigrid::buildInterpMatrix)
        ev = np.column_stack([rng.random(12) * 0.8 + 0.1, rng.random(12) * 0.8 + 0.1, np.zeros(12)])
        nb2 = np.array([og.k_nearest(e, False, False, ss) for e in ev], dtype=np.int32)
        wi = _capi.rbf_weights(2, deg, 3.0, og.points, ev, nb2, [4])
        for k, e in enumerate(ev):
            ref = og.point_interp_weights(e, deg)[0][:ss]
            assert np.abs(wi[0, k] - ref).max() <= 1e-6 * np.abs(ref).max(), (deg, k)



def _brute_knn(cloud, queries, k, dim, cloud_flag=None, query_flag=None):
    """Grid::kNearestNeighbors (grid.cpp:216-260) by full scan: k smallest (distance, index), the expression of
    oracle/setup_oracle.py:217 (squares and sum rounded separately, then sqrt)."""
    out = np.full((len(queries), k), -1, dtype=np.int64)
    for e, q in enumerate(queries):
        d2 = (cloud[:, 0] - q[0]) ** 2 + (cloud[:, 1] - q[1]) ** 2
        if dim >= 3:
            d2 = d2 + (cloud[:, 2] - q[2]) ** 2
        d = np.sqrt(d2)
        idx = np.arange(len(cloud))
        if cloud_flag is not None and query_flag is not None and query_flag[e]:
            idx = idx[(cloud_flag == 0) | (d == 0.0)]
        order = np.lexsort((idx, d[idx]))[:k]
        out[e, :len(order)] = idx[order]
    return out


@pytest.mark.parametrize("dim,n,k,jitter", [(2, 3000, 15, 0.25), (2, 3000, 28, 0.0), (3, 5000, 50, 0.25), (3, 4096, 56, 0.0),
                                             (3, 3000, 120, 0.3), (2, 2000, 200, 0.25), (3, 40, 50, 0.25)])
def test_device_knn_matches_full_scan_bitwise(host, dim, n, k, jitter):
    """mmg_knn against the full scan: identical neighbour lists including the order among equal distances
    (jitter 0: a lattice, where most distances tie and the index decides; the last case asks for more
    neighbours than the cloud holds).  Queries: the cloud's own points plus points outside its bounding box."""
    from meshlessmultigridpoisson_amd import _capi
    rng = np.random.default_rng(7 + dim + k)
    m = int(round(n ** (1.0 / dim)))
    ax = np.arange(m) / (m - 1.0)
    mesh = np.stack(np.meshgrid(*([ax] * dim), indexing="ij"), axis=-1).reshape(-1, dim)
    mesh = mesh + jitter / (m - 1.0) * rng.uniform(-1, 1, mesh.shape)
    cloud = np.zeros((len(mesh), 3))
    cloud[:, :dim] = mesh[rng.permutation(len(mesh))]
    extra = np.zeros((40, 3))
    extra[:, :dim] = rng.uniform(-0.3, 1.3, (40, dim))
    sel = rng.choice(len(cloud), size=min(len(cloud), 600), replace=False)
    queries = np.concatenate([cloud[sel], extra])
    got = _capi.knn(dim, cloud, queries, k)
    want = _brute_knn(cloud, queries, k, dim)
    assert np.array_equal(got, want)
    # Neumann rule: flagged queries skip flagged candidates except at distance 0
    cflag = (rng.uniform(size=len(cloud)) < 0.3).astype(np.uint8)
    qflag = np.concatenate([cflag[sel], np.ones(20, np.uint8), np.zeros(20, np.uint8)])
    got = _capi.knn(dim, cloud, queries, k, cflag, qflag)
    want = _brute_knn(cloud, queries, k, dim, cflag, qflag)
    assert np.array_equal(got, want)


def test_device_knn_matches_setup_oracle_and_clustered_cloud(host):
    """(a) directly against oracle/setup_oracle.py Grid.k_nearest on a Neumann square (boundary queries ignore the
    other boundary points); (b) a strongly non-uniform cloud (a dense cluster in a sparse background: the first
    search block of a background query holds too few points, of a cluster query far too many)."""
    from meshlessmultigridpoisson_amd import _capi
    from oracle import setup_oracle as so
    pts = so.square_cloud(24, seed=5)
    bpts = [i for i, (x, y, _z) in enumerate(pts) if x == 0 or x == 1 or y == 0 or y == 1]
    og = so.Grid(pts, [so.Boundary(2, bpts, [0.0] * len(bpts))], so.make_props(3), np.zeros(len(pts) + 1))
    og.set_bc_flag(0, "neumann", [0.0] * len(bpts))
    cloud = np.zeros((og.n, 3))
    cloud[:, :2] = np.asarray(og.points)[:, :2]
    bc = (np.asarray(og.bcflags) != 0).astype(np.uint8)
    assert bc.sum() == len(bpts)
    got = _capi.knn(2, cloud, cloud, og.props.stencilSize, bc, bc)
    for i in range(og.n):
        assert list(got[i]) == og.k_nearest(og.points[i], True, og.bcflags[i] != 0, og.props.stencilSize), i
    rng = np.random.default_rng(99)
    cloud = np.zeros((6000, 3))
    cloud[:1000] = rng.uniform(0, 1, (1000, 3))
    cloud[1000:] = 0.5 + 0.01 * rng.standard_normal((5000, 3))
    q = np.concatenate([cloud[:150], cloud[1000:1150]])
    assert np.array_equal(_capi.knn(3, cloud, q, 50), _brute_knn(cloud, q, 50, 3))


@pytest.mark.parametrize("dim,n,deg", [(2, 2500, 3), (3, 3375, 3)])
def test_rbf_stencils_is_search_plus_weights_and_orders_rows(host, dim, n, deg):
    """mmg_rbf_stencils through the C-ABI: (a) its lists are mmg_knn's, its weights those of mmg_rbf_weights on these
    lists, bit for bit (same kernels, the lists just never leave the device); (b) by_column returns every row as the
    same (id, weight) pairs in ascending id -- the order of a CSR row; (c) flags follow mmg_knn's Neumann rule;
    (d) a cloud smaller than the stencil reports short rows instead of weights."""
    from meshlessmultigridpoisson_amd import _capi
    m = int(round(n ** (1.0 / dim)))
    pts = host.box_cloud(m, dim, seed=3)
    ss = host.stencil_size(deg, dim)
    ops = [0, 1, 2] + ([3] if dim == 3 else [])
    nbr, w, short = _capi.rbf_stencils(dim, deg, 3.0, ss, pts, pts, ops)
    assert short == 0
    assert np.array_equal(nbr, _capi.knn(dim, pts, pts, ss))
    assert np.array_equal(w, _capi.rbf_weights(dim, deg, 3.0, pts, pts, nbr, ops))
    nbr_c, w_c, short = _capi.rbf_stencils(dim, deg, 3.0, ss, pts, pts, ops, by_column=True)
    assert short == 0
    order = np.argsort(nbr, axis=1, kind="stable")
    assert np.array_equal(nbr_c, np.take_along_axis(nbr, order, axis=1))
    for o in range(len(ops)):
        assert np.array_equal(w_c[o], np.take_along_axis(w[o], order, axis=1))
    flag = (np.abs(pts[:, :dim] - 0.5).max(axis=1) == 0.5).astype(np.uint8)      # the boundary of the box
    nbr_f, _w, short = _capi.rbf_stencils(dim, deg, 3.0, ss, pts, pts, [0], cloud_flag=flag, eval_flag=flag)
    assert short == 0 and np.array_equal(nbr_f, _capi.knn(dim, pts, pts, ss, flag, flag))
    b = np.flatnonzero(flag)
    assert all(flag[nbr_f[i, 1:]].sum() == 0 for i in b[:50])                     # only itself among the boundary points
    few = pts[: ss - 1]
    _n, _w2, short = _capi.rbf_stencils(dim, deg, 3.0, ss, few, few, [0])
    assert short == len(few)


def test_knn_edge_cases(host):
    """Empty query set, a cloud of one point, k = 1, the largest supported k (256), k beyond it (refused), duplicate
    points (distance 0 ties resolved by index)."""
    from meshlessmultigridpoisson_amd import _capi
    rng = np.random.default_rng(0)
    cloud = np.zeros((700, 3))
    cloud[:, :2] = rng.uniform(0, 1, (700, 2))
    assert _capi.knn(2, cloud, np.zeros((0, 3)), 5).shape == (0, 5)
    one = _capi.knn(2, cloud[:1], cloud[:4], 3)
    assert np.array_equal(one, np.tile(np.array([0, -1, -1], dtype=np.int32), (4, 1)))
    assert np.array_equal(_capi.knn(2, cloud, cloud, 1)[:, 0], np.arange(700))
    assert np.array_equal(_capi.knn(2, cloud, cloud[:40], 256), _brute_knn(cloud, cloud[:40], 256, 2))
    with pytest.raises(Exception):
        _capi.knn(2, cloud, cloud[:4], 257)
    dup = np.concatenate([cloud[:50], cloud[:50], cloud[50:300]])       # every one of the first 50 points twice
    assert np.array_equal(_capi.knn(2, dup, dup[:100], 9), _brute_knn(dup, dup[:100], 9, 2))


def _reference_knn(cloud, cloud_flag, q, q_flag, k):
    """grid.cpp:216-260 restated literally for one query: distances to every point, samePoint = the LAST index at
    distance 0 (:219-226), a max-heap of k candidates in which a flagged candidate is admitted only if it is
    samePoint (:236,244); returns the k smallest (distance, index) pairs in order."""
    d = np.sqrt((cloud[:, 0] - q[0]) ** 2 + (cloud[:, 1] - q[1]) ** 2)
    zero = np.flatnonzero(d == 0.0)
    same = int(zero[-1]) if len(zero) else -1
    cand = [(d[i], i) for i in range(len(cloud)) if i == same or not (q_flag and cloud_flag[i] != 0)]
    cand.sort()
    return [i for _d, i in cand[:k]]


def test_knn_flagged_query_rule_on_coincident_boundary_points(host):
    """the reference exempts ONE zero-distance candidate from the boundary exclusion (samePoint,
    the last index at distance 0); mmg_knn keeps every flagged candidate at distance 0 (documented in mmgp.h).  On a
    cloud without coincident points the two rules are the same lists; with a boundary point stored twice the device
    list is the reference's plus the extra coincident copy (and one far neighbour fewer) -- nothing else differs."""
    from meshlessmultigridpoisson_amd import _capi
    pts = host.quasi_uniform_square_cloud(12)
    flag = ((pts[:, 0] == 0) | (pts[:, 0] == 1) | (pts[:, 1] == 0) | (pts[:, 1] == 1)).astype(np.uint8)
    k = 25
    got = _capi.knn(2, pts, pts, k, flag, flag)
    for i in range(len(pts)):
        assert list(got[i]) == _reference_knn(pts, flag, pts[i], flag[i], k), i
    b = int(np.flatnonzero(flag)[5])
    dup = np.concatenate([pts, pts[b:b + 1]])           # boundary point b stored a second time (index n)
    dflag = np.append(flag, 1).astype(np.uint8)
    got = _capi.knn(2, dup, dup, k, dflag, dflag)
    n = len(pts)
    for i in range(len(dup)):
        ref = _reference_knn(dup, dflag, dup[i], dflag[i], k)
        if i in (b, n):                                  # the coincident pair: both copies at distance 0 are kept
            assert list(got[i][:2]) == [b, n] and ref[0] == n
            assert list(got[i][2:]) == ref[1:k - 1]
        else:
            assert list(got[i]) == ref, i


def test_fracstep_operator_cache_follows_polydeg_changes(host):
    """the operators of one device batch (D_x, D_y, lap from ONE factorisation per point) are cached until
    fetched; the cache is keyed on everything the stencils depend on.  D_x built at polyDeg 3 (K = 25), polyDeg then
    set to 4: the next build_derivY_mat must deliver K = 37 rows -- the reference rebuilds each operator from the
    current state (fractionalStepGrid.cpp:60-100)."""
    import ctypes
    host.set_option("device_setup", 1)
    try:
        g = host.FracStepGrid.create(host.quasi_uniform_square_cloud(41), polydeg=3, ordering=host.ORDER_MC, tile_points=128)
        out = (ctypes.c_int * 2)()
        f = host.lib().mmgh_fs_rebuild_after_polydeg_change
        f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        assert f(g.h, 4, out) == 0, host._err()
    finally:
        host.set_option("device_setup", -1)
    assert (out[0], out[1]) == (host.stencil_size(3), host.stencil_size(4))


@pytest.mark.gpu
@pytest.mark.parametrize("dim,deg,ss,ops", [(3, 3, 50, [0]), (3, 3, 50, [1, 2, 3, 0]), (2, 3, 25, [0, 1, 2]), (2, 4, 37, [4]),
                                            (2, 5, 51, [0]), (2, 5, 52, [0, 1]), (3, 2, 30, [0, 4]), (2, 6, 70, [0])])
def test_rbf_register_kernel_matches_lds_kernel_and_reproduces_polynomials(host, dim, deg, ss, ops):
    """The two kernels behind mmg_rbf_weights (rbf_setup.hip): systems of at most 72 x 72 are factorised in the
    registers of one wavefront (Gauss-Jordan, full pivoting without data movement), larger ones in LDS (LU with full
    pivoting as the reference's fullPivLu, grid.cpp:304-424, :687-712).  Same stencils through both (option
    "rbf_kernel"): weights equal to 1e-6 of the row's largest (conditioning of the scaled saddle systems; observed
    2e-7), and -- independent of either -- every row reproduces the operator on all monomials up to polyDeg, which
    is what the polynomial block of the saddle system enforces (1e-8 relative to the row's magnitude).  (2, 5, 52) is the
    reference's degree-5 stencil (73 x 73: two wavefronts, 5 x 11 values per lane), (2, 5, 51) is
    the largest system two wavefronts share at two per SIMD (72 x 72), (2, 6, 70) is 98 x 98 -- the reference's live
    fine polyDeg 6 (FractionalStepSim.cpp:201-203): two wavefronts with 7 x 14 values per lane."""
    from meshlessmultigridpoisson_amd import _capi
    rng = np.random.default_rng(11)
    side = 40 if dim == 2 else 14
    ax = [np.arange(side) / (side - 1.0)] * dim
    pts = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, dim)
    pts = pts + (rng.random(pts.shape) - 0.5) * 0.4 / (side - 1.0)
    xyz = np.zeros((len(pts), 3))
    xyz[:, :dim] = pts
    ev = xyz[rng.choice(len(xyz), 1500, replace=False)] + 0.2 / (side - 1.0) * (rng.random((1500, 3)) - 0.5) * (np.arange(3) < dim)
    nbr = _capi.knn(dim, xyz, ev, ss)
    try:
        _capi.set_option("rbf_kernel", 1)
        w_lds = _capi.rbf_weights(dim, deg, 3.0, xyz, ev, nbr, ops)
    finally:
        _capi.set_option("rbf_kernel", 0)
    w = _capi.rbf_weights(dim, deg, 3.0, xyz, ev, nbr, ops)
    assert np.isfinite(w).all()
    scale = np.abs(w_lds).max(axis=2, keepdims=True)
    assert (np.abs(w - w_lds) / scale).max() <= 1e-6
    try:   # systems of 57..72 unknowns: two wavefronts per stencil by default, one with option 2 (same pivots)
        _capi.set_option("rbf_kernel", 2)
        w_one = _capi.rbf_weights(dim, deg, 3.0, xyz, ev, nbr, ops)
    finally:
        _capi.set_option("rbf_kernel", 0)
    assert (np.abs(w_one - w_lds) / scale).max() <= 1e-6
    # polynomial reproduction in coordinates centred on the evaluation point (so the check does not cancel)
    rel = xyz[nbr] - ev[:, None, :]                                   # [n_eval][ss][3]
    h = np.abs(rel).max(axis=(1, 2))                                   # stencil radius
    for o, op in enumerate(ops):
        for (a, b, c) in _monomials(dim, deg):
            p = rel[..., 0] ** a * rel[..., 1] ** b * rel[..., 2] ** c
            got = (w[o] * p).sum(axis=1)
            if op == 4:
                want = 1.0 if a + b + c == 0 else 0.0
            elif op == 0:
                want = 2.0 if sorted((a, b, c)) == [0, 0, 2] else 0.0
            else:
                e = [0, 0, 0]
                e[op - 1] = 1
                want = 1.0 if [a, b, c] == e else 0.0
            mag = (np.abs(w[o]) * np.abs(p)).sum(axis=1) + h ** (a + b + c) * 1e-300
            assert (np.abs(got - want) / np.maximum(mag, abs(want))).max() <= 1e-8, (op, a, b, c)


@pytest.mark.gpu
def test_rbf_register_kernel_on_coincident_stencil_points(host):
    """Two coincident stencil points make the saddle system singular (two equal rows and columns): elimination stops
    at the rank (no positive pivot candidate left), the unknowns never pivoted on stay zero -- finite weights from both
    kernels, no fault; the solution is not unique, so nothing more is compared."""
    from meshlessmultigridpoisson_amd import _capi
    rng = np.random.default_rng(5)
    xyz = np.zeros((400, 3))
    xyz[:, :2] = rng.random((400, 2))
    xyz[1] = xyz[0]
    ev = xyz[:50].copy()
    nbr = np.tile(np.arange(25, dtype=np.int32), (50, 1))              # every stencil holds the coincident pair
    for mode in (0, 1, 2):
        try:
            _capi.set_option("rbf_kernel", mode)
            w = _capi.rbf_weights(2, 3, 3.0, xyz, ev, nbr, [0, 4])
        finally:
            _capi.set_option("rbf_kernel", 0)
        assert np.isfinite(w).all()
    # the same in 3-D: 70 x 70, the shape two wavefronts share (both must leave the elimination at the same step)
    xyz3 = rng.random((400, 3))
    xyz3[1] = xyz3[0]
    nbr3 = np.tile(np.arange(50, dtype=np.int32), (50, 1))
    for mode in (0, 1, 2):
        try:
            _capi.set_option("rbf_kernel", mode)
            w = _capi.rbf_weights(3, 3, 3.0, xyz3, xyz3[:50].copy(), nbr3, [0, 1, 2, 3])
        finally:
            _capi.set_option("rbf_kernel", 0)
        assert np.isfinite(w).all()
