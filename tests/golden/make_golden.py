"""
make_golden.py -- generates the committed golden fixtures tests/golden/*.npz.

Inputs (small seeded 2-D clouds, RBF-FD Laplacians, transfer matrices) come from
oracle/setup_oracle.py (numpy restatement of the reference's setup); expected
outputs come from oracle/mmg_oracle.c (plain-C restatement of the reference's
V-cycle hot path).  The reference itself cannot run here (Eigen absent) and
ships no fixtures, so these vectors pin the ORACLE, not the reference:
"parity unpinned" (DESIGN.md).  The ref_utils fixture, in contrast, IS produced
by the reference's own Eigen-free sources compiled into oracle/_ref.

Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_c as oc  # noqa: E402
from oracle import setup_oracle as so  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NCYC = 20


def pack_case(name, grids, R, P, frac_step=False):
    d = {"nlevels": np.int32(len(grids)), "frac_step": np.int32(frac_step)}
    levels = [oc.Level.from_grid(g) for g in grids]
    for i, (g, lv) in enumerate(zip(grids, levels)):
        p = f"L{i}_"
        d[p + "points"] = g.points
        d[p + "rowptr"], d[p + "col"], d[p + "val"] = lv.rowptr, lv.col, lv.val
        d[p + "bcflags"] = lv.bcflags
        d[p + "x0"], d[p + "b0"] = lv.x.copy(), lv.b.copy()
        d[p + "btype"], d[p + "bptr"], d[p + "bpts"], d[p + "bvals"] = lv.btype, lv.bptr, lv.bpts, lv.bvals
        d[p + "meta"] = np.array([lv.n, lv.a_size, lv.neumann, lv.iters], dtype=np.int32)
        d[p + "omega"] = np.float64(lv.omega)
        d[p + "polydeg"] = np.int32(g.props.polyDeg)
    for i in range(len(grids)):
        for nm, M in (("R", R[i]), ("P", P[i])):
            if M is None:
                continue
            p = f"{nm}{i}_"
            d[p + "shape"] = np.array([M["rows"], M["cols"]], dtype=np.int32)
            d[p + "colptr"], d[p + "rowidx"], d[p + "val"] = M["colptr"], M["rowidx"], M["val"]

    # ---- expected outputs from the C oracle --------------------------------
    fine = levels[-1]
    probe = oc.Level(fine.n, fine.rowptr, fine.col, fine.val, fine.x, fine.b, fine.bcflags, fine.neumann,
                     fine.omega, fine.iters, fine.btype, fine.bptr, fine.bpts, fine.bvals)
    probe.boundary_op(0)
    d["fine_resid0"] = probe.residual()
    probe.sor_sweeps(1)
    d["fine_x_after_1sweep"] = probe.x.copy()
    probe.sor_sweeps(probe.iters - 1)
    d["fine_x_after_sor"] = probe.x.copy()
    d["fine_resid_after_sor"] = probe.residual()
    d["fine_ratio_after_sor"] = np.float64(probe.residual_ratio())

    mg = oc.Multigrid(levels, [oc.Transfer.from_dict(r) if r else None for r in R],
                      [oc.Transfer.from_dict(p) if p else None for p in P], frac_step=frac_step)
    for _ in range(NCYC):
        mg.vcycle()
    d["resid_history"] = np.array(mg.residuals)
    d["final_ratio"] = np.float64(mg.residual())
    for i, lv in enumerate(levels):
        d[f"L{i}_x_final"] = lv.x.copy()
        d[f"L{i}_b_final"] = lv.b.copy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "residuals", " ".join(f"{r:.3e}" for r in mg.residuals[:: max(1, NCYC // 6)]))


def build(sizes, polys, neumann, bvalue_fn=None, seed=12345):
    grids = []
    for i, (ns, pd) in enumerate(zip(sizes, polys)):
        pts = so.square_cloud(ns, seed=seed + i)
        props = so.make_props(pd)
        if neumann:
            g = so.gen_grid_neumann_square(pts, props, coarse=(i != len(sizes) - 1))
        else:
            g = so.gen_grid_dirichlet_square(pts, props, bvalue_fn=bvalue_fn)
        grids.append(g)
    R, P = so.build_matrices(grids)
    return grids, R, P


def build_live(sizes, polys):
    """The reference's live parameter set (run_tests / run_frac_step_test: Neumann, fine polyDeg 4-6, coarse 3) on
    Gmsh-like clouds (setup_oracle.quasi_uniform_square_cloud), the reference's RCM order."""
    grids = []
    for i, (ns, pd) in enumerate(zip(sizes, polys)):
        pts = so.quasi_uniform_square_cloud(ns)
        grids.append(so.gen_grid_neumann_square(pts, so.make_props(pd), coarse=(i != len(sizes) - 1)))
    R, P = so.build_matrices(grids)
    return grids, R, P


def ref_utils_fixture():
    """Outputs of the REFERENCE's own leaf functions (oracle/_ref)."""
    L = oc.ref_lib()
    if L is None:
        print("oracle/_ref not built; skipping ref_utils fixture")
        return
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rng = np.random.default_rng(2024)
    d = {}
    # distance
    pq = rng.random((64, 2, 3))
    d["dist_in"] = pq
    d["dist_out"] = np.array([L.ref_distance(p[0].ctypes.data_as(dp), p[1].ctypes.data_as(dp)) for p in pq])
    # shifting_scaling
    pts = rng.random((25, 3))
    ev = rng.random(3)
    out = np.zeros((27, 3))
    L.ref_shifting_scaling(pts.ctypes.data_as(dp), 25, ev.ctypes.data_as(dp), out.ctypes.data_as(dp))
    d["ss_pts"], d["ss_eval"], d["ss_out"] = pts, ev, out
    # RCM on a kNN graph
    cloud = so.square_cloud(9, seed=5)
    g = so.Grid(cloud, [so.Boundary(1, [], [])], so.make_props(3), np.zeros(len(cloud)))
    adj = [g.k_nearest(cloud[i], False, False, 9) for i in range(len(cloud))]
    ptr = np.zeros(len(adj) + 1, dtype=np.int32)
    ptr[1:] = np.cumsum([len(a) for a in adj])
    idx = np.array([j for a in adj for j in a], dtype=np.int32)
    order = np.zeros(len(adj), dtype=np.int32)
    n = L.ref_rcm(ptr.ctypes.data_as(ip), idx.ctypes.data_as(ip), len(adj), order.ctypes.data_as(ip))
    d["rcm_ptr"], d["rcm_idx"], d["rcm_order"] = ptr, idx, order[:n]
    # .msh reader on a small MSH 2.2 file written here
    msh = os.path.join(OUT, "tiny_square.msh")
    cloud4 = so.square_cloud(4, seed=9)
    tris = []
    for j in range(3):
        for i in range(3):
            a = j * 4 + i
            tris.append((a, a + 1, a + 5))
            tris.append((a, a + 5, a + 4))
    with open(msh, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(cloud4))
        for i, p in enumerate(cloud4):
            f.write("%d %.17g %.17g %.17g\n" % (i + 1, p[0], p[1], p[2]))
        f.write("$EndNodes\n$Elements\n%d\n" % (len(tris) + 2))
        f.write("1 15 2 0 1 1\n")
        f.write("2 1 2 0 1 1 2\n")
        for k, t in enumerate(tris):
            f.write("%d 2 2 0 1 %d %d %d\n" % (k + 3, t[0] + 1, t[1] + 1, t[2] + 1))
        f.write("$EndElements\n")
    xyz = np.zeros((len(cloud4), 3))
    n = L.ref_points_from_msh(msh.encode(), xyz.ctypes.data_as(dp), len(cloud4))
    assert n == len(cloud4)
    d["msh_points"] = xyz
    flags = np.array([1 if (p[0] in (0, 1) or p[1] in (0, 1)) else 0 for p in cloud4], dtype=np.int32)
    conn = np.zeros((len(cloud4), 2), dtype=np.int32)
    L.ref_bound_pts_conn(msh.encode(), flags.ctypes.data_as(ip), len(cloud4), conn.ctypes.data_as(ip))
    d["msh_bcflags"], d["msh_conn"] = flags, conn
    # writeVectorToTxt formatting (default ostream precision)
    v = np.array([1.0, 0.1234567891234, 1e-12, 123456789.0, -2.5e10])
    txt = os.path.join(OUT, "ref_vector.txt")
    L.ref_write_vector_txt(v.ctypes.data_as(dp), len(v), txt.encode())
    d["txt_vec"] = v
    np.savez_compressed(os.path.join(OUT, "ref_utils.npz"), **d)
    print("ref_utils fixture written")


if __name__ == "__main__":
    oc.build()
    ref_utils_fixture()
    g, R, P = build([13, 25, 49], [3, 3, 4], neumann=False)
    pack_case("dirichlet_3level", g, R, P)
    g, R, P = build([25, 49], [3, 3], neumann=True)
    pack_case("neumann_2level", g, R, P)
    g, R, P = build([13, 25, 49], [3, 3, 3], neumann=True)
    pack_case("neumann_3level", g, R, P)
    g, R, P = build([13, 25], [3, 3], neumann=False, bvalue_fn=lambda x, y: 1.0 + x + 2 * y)
    pack_case("dirichlet_2level_inhomog", g, R, P)
    if "--live" in sys.argv or not os.path.exists(os.path.join(OUT, "neumann_live_L6_3level.npz")):
        g, R, P = build_live([13, 25, 49], [3, 3, 6])       # O(N^2) python setup with 98 x 98 solves: a few minutes
        pack_case("neumann_live_L6_3level", g, R, P)
