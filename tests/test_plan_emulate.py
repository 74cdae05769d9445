"""CPU check of the packed tile plan (the bytes the gfx950 kernels stream): the
adversarially-concurrent interpreter in tests/support/plan_emulate.cpp must
reproduce the oracle's sequential Gauss-Seidel iterates for ANY point ordering,
tile size and lanes-per-row.  Tolerance: 1e-12 relative (only the association
order inside one row's dot product differs)."""
import numpy as np
import pytest

from tests import helpers as H

CASES = ["dirichlet_3level", "neumann_2level", "neumann_3level", "dirichlet_2level_inhomog", "neumann_live_L6_3level"]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tile,L", [(64, 4), (256, 2), (100, 16), (512, 1), (37, 8)])
def test_sweeps_match_oracle(name, tile, L):
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    e = H.EmuLevel(la, tile_size=tile, lanes_per_row=L)
    o.boundary_op(0)
    e.x[:] = o.x
    o.sor_sweeps(3)
    e.sweeps(3)
    assert H.rel_err(e.x, o.x) < 1e-12
    r, nrm = e.residual()
    ro = o.residual()
    assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(o.b).max())
    assert abs(nrm - np.abs(ro).sum()) <= 1e-10 * np.abs(ro).sum() + 1e-12


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tile,L,waves", [(64, 8, 4), (128, 16, 4), (96, 8, 2), (200, 8, 6), (48, 16, 3)])
def test_dense_multiwave_layout_matches_oracle(name, tile, L, waves):
    """Dense plans (plan.hpp): fixed-shape groups, rows list-scheduled into rounds of `waves` groups.  The
    interpreter lets ALL rows of a round read before any of them writes (the wavefronts of a workgroup run a
    round concurrently) -- sweeps and residuals must still be the oracle's sequential ones."""
    import ctypes
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    o = H.oracle_level(la)
    e = H.EmuLevel(la, tile_size=tile, lanes_per_row=L, waves_per_tile=waves)
    lib = H.emu_lib()
    lib.emu_level_waves.argtypes = [ctypes.c_void_p]
    rowlen = int(np.diff(la["rowptr"])[:-1 if la["neumann"] else None][la["bcflags"] == 0].max())
    if rowlen - 2 <= 8 * L:
        assert lib.emu_level_waves(e.h) == waves
    else:   # rows beyond a dense row slot (the polyDeg-6 Neumann fixture: ~190 entries): multi-slot rows with 4 / 6
        assert lib.emu_level_waves(e.h) in (4, 6, 0)   # wavefronts (Plan::dense_long), or the packed stream
    assert lib.emu_level_slot_bits(e.h) == 16 or lib.emu_level_waves(e.h) == 0
    o.boundary_op(0)
    e.x[:] = o.x
    o.sor_sweeps(3)
    e.sweeps(3)
    assert H.rel_err(e.x, o.x) < 1e-12
    r, nrm = e.residual()
    ro = o.residual()
    assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(o.b).max())
    assert abs(nrm - np.abs(ro).sum()) <= 1e-10 * np.abs(ro).sum() + 1e-12
    assert lib.emu_last_error() in (b"", None) or b"dense group head" not in lib.emu_last_error()


def test_dense_layout_falls_back_for_long_rows():
    """Rows of more than 8 entries per lane do not fit one dense row slot: up to 256 entries they take several row slots
    (Plan::dense_long, 150 entries here: three slots of 16 lanes x 4 entries); beyond that (270 entries) the level keeps
    the packed stream.  Results unchanged either way.  A request for 4 lanes per row is served with the dense layout's
    own choice (8 or 16 lanes)."""
    rng = np.random.default_rng(2)
    for n, k, want_waves in ((220, 150, 4), (300, 270, 0)):
        _long_rows_case(rng, n, k, want_waves)
    lib = H.emu_lib()
    case = H.load_case("dirichlet_3level")
    e4 = H.EmuLevel(H.level_arrays(case, 2), tile_size=64, lanes_per_row=4, waves_per_tile=4)   # K = 37
    assert lib.emu_level_waves(e4.h) == 4


def _long_rows_case(rng, n, k, want_waves):
    rowptr, col, val = [0], [], []
    for i in range(n):
        c = np.sort(rng.choice(np.delete(np.arange(n), i), size=k - 1, replace=False))
        c = np.sort(np.append(c, i))
        v = -rng.random(k)
        v[c == i] = k + 1.0
        col += c.tolist()
        val += v.tolist()
        rowptr.append(len(col))
    la = dict(n=n, a_size=n, rowptr=np.array(rowptr, dtype=np.int32), col=np.array(col, dtype=np.int32), val=np.array(val),
              bcflags=np.zeros(n, dtype=np.int32), neumann=0, omega=1.2, iters=2, btype=np.zeros(0, dtype=np.int32),
              bptr=np.zeros(1, dtype=np.int32), bpts=np.zeros(0, dtype=np.int32), bvals=np.zeros(0),
              x0=np.zeros(n), b0=rng.standard_normal(n))
    e = H.EmuLevel(la, tile_size=64, lanes_per_row=16, waves_per_tile=4)
    assert e.waves() == want_waves and e.dense_long() == (want_waves > 0)
    o = H.oracle_level(la)
    o.sor_sweeps(2)
    e.sweeps(2)
    assert H.rel_err(e.x, o.x) < 1e-12


@pytest.mark.parametrize("name", ["neumann_2level", "neumann_3level"])
def test_bound_eval_matches_oracle(name):
    case = H.load_case(name)
    la = H.level_arrays(case, case["nlevels"] - 1)
    rng = np.random.default_rng(3)
    la["x0"] = rng.standard_normal(la["a_size"])
    o = H.oracle_level(la)
    e = H.EmuLevel(la, tile_size=128, lanes_per_row=4)
    o.bound_eval_neumann()
    e.bound_eval()
    assert H.rel_err(e.x, o.x) < 1e-12


def test_schedule_is_exact_for_storage_order_and_reports_phases():
    """RCM ordering (what the fixtures use) gives long dependency chains: many
    phases, still exact.  Every interior row appears exactly once."""
    case = H.load_case("dirichlet_3level")
    la = H.level_arrays(case, 2)
    e = H.EmuLevel(la, tile_size=64, lanes_per_row=4)
    info = e.info()
    assert info["n_tiles"] == (la["n"] + 63) // 64
    assert info["n_phases"] >= 2
    assert info["max_slots"] <= 7680


@pytest.mark.parametrize("name", ["dirichlet_3level", "neumann_2level"])
def test_transfers_match_oracle(name):
    from oracle import oracle_c as oc
    case = H.load_case(name)
    rng = np.random.default_rng(0)
    for key in ("R1", "P0"):
        shape = case[key + "_shape"]
        t = oc.Transfer(*shape, case[key + "_colptr"], case[key + "_rowidx"], case[key + "_val"])
        x = rng.standard_normal(int(shape[1]))
        y0 = rng.standard_normal(int(shape[0]))
        assert H.rel_err(H.emu_transfer_apply(shape, case[key + "_colptr"], case[key + "_rowidx"], case[key + "_val"], x),
                         t.apply(x)) < 1e-13
        ya = H.emu_transfer_apply(shape, case[key + "_colptr"], case[key + "_rowidx"], case[key + "_val"], x, add_to=y0, L=8)
        assert H.rel_err(ya, y0 + t.apply(x)) < 1e-13


def test_explicit_zeros_are_dropped_but_multiplier_kept():
    case = H.load_case("neumann_2level")
    la = H.level_arrays(case, 1)
    e = H.EmuLevel(la, tile_size=128, lanes_per_row=4)
    nnz = e.L.emu_level_nnz(e.h)
    rowptr, col, val, bc, n = la["rowptr"], la["col"], la["val"], la["bcflags"], la["n"]
    expect = 0
    for i in range(n):
        if bc[i] != 0:
            continue
        c = col[rowptr[i]:rowptr[i + 1]]
        v = val[rowptr[i]:rowptr[i + 1]]
        expect += int(((c != i) & (c != n) & (v != 0.0)).sum())
    assert nnz == expect


def test_twelve_bit_slot_stream_and_sixteen_bit_option():
    """Level plans with L = 2 ... 16 and <= 4096 LDS slots per tile pack the tile-local column indices in 12 bits
    (plan.hpp: slot_words; the default); mmg_set_option("slot_bits", 16) keeps 16-bit slots.  Same rows, same
    entries, fewer stream bytes, same arithmetic."""
    import ctypes
    case = H.load_case("dirichlet_3level")
    la = H.level_arrays(case, case["nlevels"] - 1)
    L = H.emu_lib()
    L.emu_level_slot_bits.argtypes = [ctypes.c_void_p]
    e12 = H.EmuLevel(la, tile_size=64, lanes_per_row=2)
    assert L.emu_level_slot_bits(e12.h) == 12
    e12n = H.EmuLevel(H.level_arrays(H.load_case("neumann_3level"), 2), tile_size=48, lanes_per_row=4)
    assert L.emu_level_slot_bits(e12n.h) == 12
    e8 = H.EmuLevel(la, tile_size=64, lanes_per_row=8)
    assert L.emu_level_slot_bits(e8.h) == 12
    e1 = H.EmuLevel(la, tile_size=64, lanes_per_row=1)       # one lane per row keeps 16-bit slots
    assert L.emu_level_slot_bits(e1.h) == 16
    L.emu_set_slot_bits(16)
    try:
        e16 = H.EmuLevel(la, tile_size=64, lanes_per_row=2)
        assert L.emu_level_slot_bits(e16.h) == 16
    finally:
        L.emu_set_slot_bits(12)
    assert L.emu_level_nnz(e12.h) == L.emu_level_nnz(e16.h)
    assert L.emu_level_stream_bytes(e12.h) < L.emu_level_stream_bytes(e16.h)
    e12.sweeps(2)
    e16.sweeps(2)
    assert np.array_equal(e12.x, e16.x)          # the index width changes no arithmetic
    o = H.oracle_level(H.level_arrays(H.load_case("neumann_3level"), 2))
    o.sor_sweeps(2)
    e12n.sweeps(2)
    assert H.rel_err(e12n.x, o.x) < 1e-12


def test_threaded_csc_to_csr_keeps_eigens_accumulation_order():
    """libmmgp's csc_to_csr (mmg_transfer_create for the reference's column-major transfers) runs on all host
    threads from 1e5 non-zeros on: every thread owns a range of columns and writes behind the threads before it.
    The product through the packed gather plan must equal scipy's, and -- the rows being summed in ascending
    column order, Eigen's accumulation order -- be bitwise the sequential row sums."""
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    rows, cols, per_col = 6000, 5000, 30     # rows stay below 64 entries: one lane sums a whole row
    rowidx = np.concatenate([np.sort(rng.choice(rows, per_col, replace=False)) for _ in range(cols)]).astype(np.int32)
    colptr = (np.arange(cols + 1) * per_col).astype(np.int32)
    val = rng.standard_normal(len(rowidx))
    x = rng.standard_normal(cols)
    assert len(rowidx) >= 100000
    y = H.emu_transfer_apply((rows, cols), colptr, rowidx, val, x, L=1)   # one lane per row: plain sequential sums
    M = sp.csc_matrix((val, rowidx, colptr), shape=(rows, cols)).tocsr()
    M.sort_indices()
    assert np.diff(M.indptr).max() < 64
    want = M @ x
    from fractions import Fraction
    seq = np.zeros(rows)
    for i in range(rows):           # s = fma(a, x, s) entry by entry, like the kernels and their emulator
        s = 0.0
        for p in range(M.indptr[i], M.indptr[i + 1]):
            s = float(Fraction(float(M.data[p])) * Fraction(float(x[M.indices[p]])) + Fraction(s))
        seq[i] = s
    assert np.array_equal(y, seq)
    assert np.allclose(y, want, rtol=1e-12, atol=1e-12)


def test_dense_groups_with_rows_over_several_row_slots():
    """Plan::dense_long: the implicitly eliminated Neumann level of a 3-D hierarchy has rows of up to ~200 entries, more
    than the 128 a dense row slot of 16 lanes holds at most.  Such rows take several consecutive row slots of one group
    (continuation slots: gid = kNoRow, self = kContSlot), the head slot adds their sums.  The adversarial interpreter of
    the packed bytes follows the oracle through sweeps and residual, with 4 and with 6 wavefronts per tile."""
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("device_setup", 0)
    pts = host.box_cloud(15, 3, seed=11, edges=False)
    g = host.Multigrid([pts], [3], dim=3, neumann=True, ordering=host.ORDER_MC, tile_points=128).grid(0)
    la = g.level_arrays()
    n = la["n"]
    rowlen = np.diff(la["rowptr"])[:n]
    assert rowlen.max() > 130, rowlen.max()                      # longer than any single dense row slot
    rng = np.random.default_rng(5)
    x0 = rng.standard_normal(len(la["x0"]))
    for waves in (4, 6):
        lv = H.oracle_level(la)
        emu = H.EmuLevel(la, tile_ptr=g.tile_ptr(), waves_per_tile=waves)
        assert emu.waves() == waves and emu.dense_long()
        lv.x[:] = x0
        emu.x[:] = x0
        lv.sor_sweeps(3)
        emu.sweeps(3)
        assert H.rel_err(emu.x, lv.x) < 1e-12, waves
        r_e, _nrm = emu.residual()
        assert H.rel_err(r_e, lv.residual()) < 1e-11, waves


def test_sweep_ordered_level_dense_layout_with_one_wavefront_per_tile():
    """Round 3: a 2-D level ordered by Grid::mc_order_points as a lexicographic SWEEP inside its tiles (point order 2,
    the order in which the reference's over-relaxed cycle converges) has ~4 uncoupled rows per dependency level.  The
    dense layout with ONE wavefront per tile (waves_per_tile = -1: rounds of a single group) streams half the bytes of
    rounds of two groups, and sweeps / residual are the oracle's (interpreter: all rows of a round read first)."""
    import ctypes
    from meshlessmultigridpoisson_amd import _host as host
    host.set_option("point_colouring", 2)
    try:
        g = host.Grid.create_square(host.quasi_uniform_square_cloud(61), 4, kind=host.KIND_DIRICHLET, ordering=host.ORDER_MC,
                                    tile_points=256)
    finally:
        host.set_option("point_colouring", -1)
    la = g.level_arrays()
    rng = np.random.default_rng(4)
    la["x0"] = rng.standard_normal(la["a_size"])
    lib = H.emu_lib()
    lib.emu_level_waves.argtypes = [ctypes.c_void_p]
    o = H.oracle_level(la)
    o.sor_sweeps(2)
    ro = o.residual()
    nbytes = {}
    for waves, lanes in ((-1, 8), (2, 8), (-1, 16)):
        e = H.EmuLevel(la, tile_ptr=g.tile_ptr(), lanes_per_row=lanes, waves_per_tile=waves)
        assert lib.emu_level_waves(e.h) == waves
        e.sweeps(2)
        assert H.rel_err(e.x, o.x) < 1e-12
        r, _nrm = e.residual()
        assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(ro).max())
        nbytes[(waves, lanes)] = lib.emu_level_stream_bytes(e.h)
        # the sweep order shows as a long dependency chain: several times the ~20 colour classes of a coloured tile
        assert e.info()["n_groups"] / e.info()["n_tiles"] > 40 * (2 if waves == 2 else 1)
    assert nbytes[(-1, 8)] < 0.6 * nbytes[(2, 8)]
    # K = 37 on 16 lanes x 3 entries: rounds of 4 rows are nearly full where rounds of 8 are not
    assert nbytes[(-1, 16)] < 0.9 * nbytes[(-1, 8)]


def test_dense_layout_extra_entry_plane_for_3d_stencils():
    """Round 3: rows of 49 off-diagonal entries (3-D, K = 50) in dense groups of 16 lanes x 3 entries + ONE extra entry
    per row (value after the slot section, slot in RowMeta::flags >> 1) instead of 16 x 4 with 15 empty slots.
    mmg_set_option("dense_xtra", 0) keeps the old shape; both reproduce the oracle."""
    import ctypes
    from meshlessmultigridpoisson_amd import _host as host
    g = host.Grid.create_square(host.box_cloud(16, 3, seed=3), 3, dim=3, kind=host.KIND_DIRICHLET, ordering=host.ORDER_MC,
                                tile_points=256)
    la = g.level_arrays()
    assert int(np.diff(la["rowptr"]).max()) == 50
    rng = np.random.default_rng(8)
    la["x0"] = rng.standard_normal(la["a_size"])
    o = H.oracle_level(la)
    o.sor_sweeps(2)
    ro = o.residual()
    lib = H.emu_lib()
    lib.emu_level_dense_xtra.argtypes = [ctypes.c_void_p]
    lib.emu_set_dense_xtra(2)      # 1 (default): only plans of >= 2e6 rows, where the bytes matter; 2: always
    try:
        e = H.EmuLevel(la, tile_ptr=g.tile_ptr(), lanes_per_row=16, waves_per_tile=4)
    finally:
        lib.emu_set_dense_xtra(1)
    assert lib.emu_level_dense_xtra(e.h) == 1
    e.sweeps(2)
    assert H.rel_err(e.x, o.x) < 1e-12
    r, nrm = e.residual()
    assert np.abs(r - ro).max() <= 1e-11 * max(1.0, np.abs(ro).max())
    assert abs(nrm - np.abs(ro).sum()) <= 1e-10 * np.abs(ro).sum()
    assert lib.emu_last_error() in (b"", None)


@pytest.mark.parametrize("n,max_len,dfrac", [(1, 1, 0.0), (2, 2, 0.0), (65, 3, 0.0), (300, 40, 0.2), (257, 200, 0.0), (120, 20, 1.0)])
@pytest.mark.parametrize("waves", [1, 4])
def test_edge_case_levels_ragged_rows_single_point_all_boundary(n, max_len, dfrac, waves):
    """Shapes no stencil generator produces (empty / ragged / extreme inputs): a level of ONE point, of two,
    rows of 1 ... 200 entries side by side (diagonal-only rows included), 20 % Dirichlet points, a level where EVERY point
    is a Dirichlet point (no row is ever relaxed; the masked residual is zero).  Packed and dense plans are built and
    the interpreter reproduces the oracle."""
    la = H.ragged_level(n, seed=n + max_len, max_len=max_len, dirichlet_frac=dfrac)
    o = H.oracle_level(la)
    e = H.EmuLevel(la, tile_size=64, lanes_per_row=0, waves_per_tile=waves)
    o.boundary_op(0)
    e.x[:] = o.x
    o.sor_sweeps(3)
    e.sweeps(3)
    assert H.rel_err(e.x, o.x) < 1e-12
    r, nrm = e.residual()
    ro = o.residual()
    # the interpreter leaves the Dirichlet mask to the caller (the device kernel scatters zeros): compare interior rows
    inter = la["bcflags"] == 0
    if inter.any():
        assert np.abs(r[inter] - ro[inter]).max() <= 1e-11 * max(1.0, np.abs(ro).max())
    else:
        assert np.all(ro == 0.0) and np.array_equal(e.x, o.x)
