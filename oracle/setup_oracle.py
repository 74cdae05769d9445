"""
setup_oracle.py -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE).

numpy restatement of the *setup* half of michaelxu3/MeshlessMultigridPoisson:
everything that produces the inputs of the V-cycle hot path (stencil search,
PHS+polynomial RBF-FD weights, Laplacian assembly with Neumann rows / multiplier
row / implicit boundary elimination, RBF interpolation transfer matrices, RCM
ordering) and the manufactured-problem grid factories of testing_functions.cpp.
Small clouds only (O(N^2) neighbour search, exactly like the reference).

PARITY UNPINNED: the reference ships no fixtures and cannot be built here
(Eigen absent); every function cites the reference lines it follows
(paths relative to /root/reference/MeshlessPoisson/).  The Eigen-free leaf
functions (distance, shifting_scaling, RCM, .msh reader) ARE pinned against the
reference's own sources compiled into oracle/_ref (tests/test_ref_utils.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
from __future__ import annotations

import math
from collections import deque
from dataclasses import dataclass, field

import numpy as np

PI = 3.141592653589793238462643383279  # testing_functions.hpp:9


# --------------------------------------------------------------------------
# leaf utilities (general_computation_functions.cpp)
# --------------------------------------------------------------------------
def distance(p, q):
    """general_computation_functions.cpp:4-6 -- 2-D distance, z ignored."""
    return math.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2)


def shifting_scaling(pts, eval_pt):
    """general_computation_functions.cpp:82-107.

    Returns scaled stencil points, then the (scale,scale,scale) marker, then the
    scaled evaluation point -- the reference's list layout."""
    pts = np.asarray(pts, dtype=np.float64)
    min_x, max_x = pts[:, 0].min(), pts[:, 0].max()
    min_y, max_y = pts[:, 1].min(), pts[:, 1].max()
    scale = max(max_x - min_x, max_y - min_y)
    out = np.zeros((len(pts) + 2, 3))
    out[: len(pts), 0] = (pts[:, 0] - min_x) / scale
    out[: len(pts), 1] = (pts[:, 1] - min_y) / scale
    out[len(pts)] = (scale, scale, scale)
    out[len(pts) + 1, 0] = (eval_pt[0] - min_x) / scale
    out[len(pts) + 1, 1] = (eval_pt[1] - min_y) / scale
    return out


def cuthill_mckee_ordering(adjacency):
    """general_computation_functions.cpp:108-130 -- plain BFS from node 0 in
    adjacency-list order (no degree sorting)."""
    n = len(adjacency)
    visited = [False] * n
    order = []
    q = deque([0])
    visited[0] = True
    while q:
        cur = q.popleft()
        order.append(cur)
        for a in adjacency[cur]:
            if not visited[a]:
                visited[a] = True
                q.append(a)
    return order


def reverse_cuthill_mckee_ordering(adjacency):
    """general_computation_functions.cpp:131-134."""
    return cuthill_mckee_ordering(adjacency)[::-1]


# --------------------------------------------------------------------------
# dense full-pivot LU (Eigen::FullPivLU::solve, used at grid.cpp:335,374,418,710)
# --------------------------------------------------------------------------
def full_piv_lu_solve(a, b):
    """Gaussian elimination with complete pivoting: at step k the pivot is the
    entry of largest magnitude of the trailing block (first one met in
    column-major order on ties), rows and columns are swapped, then P b is
    pushed through unit-lower L and upper U and the column permutation undone."""
    a = np.array(a, dtype=np.float64)
    n = a.shape[0]
    rhs = np.array(b, dtype=np.float64)
    colperm = np.arange(n)
    for k in range(n):
        sub = np.abs(a[k:, k:])
        flat = int(np.argmax(sub.T))  # column-major scan
        pc, pr = divmod(flat, n - k)
        pr += k
        pc += k
        if a[pr, pc] == 0.0:
            break
        if pr != k:
            a[[k, pr], :] = a[[pr, k], :]
            rhs[[k, pr]] = rhs[[pr, k]]
        if pc != k:
            a[:, [k, pc]] = a[:, [pc, k]]
            colperm[[k, pc]] = colperm[[pc, k]]
        if k + 1 < n:
            a[k + 1:, k] /= a[k, k]
            a[k + 1:, k + 1:] -= np.outer(a[k + 1:, k], a[k, k + 1:])
    # forward (unit lower)
    for k in range(n):
        rhs[k + 1:] -= a[k + 1:, k] * rhs[k]
    # backward (upper)
    y = np.zeros(n)
    for k in range(n - 1, -1, -1):
        y[k] = (rhs[k] - a[k, k + 1:] @ y[k + 1:]) / a[k, k]
    x = np.zeros(n)
    x[colperm] = y
    return x


# --------------------------------------------------------------------------
# sparse assembly (Eigen setFromTriplets semantics)
# --------------------------------------------------------------------------
def csr_from_triplets(n_rows, n_cols, trip):
    """Eigen::SparseMatrix<double,RowMajor>::setFromTriplets + makeCompressed:
    entries sorted by (row, col), duplicates summed in insertion order,
    explicit zeros kept (grid.cpp:590-591, 660-661)."""
    rows = np.array([t[0] for t in trip], dtype=np.int64)
    cols = np.array([t[1] for t in trip], dtype=np.int64)
    vals = np.array([t[2] for t in trip], dtype=np.float64)
    order = np.lexsort((np.arange(len(trip)), cols, rows))
    rowptr = np.zeros(n_rows + 1, dtype=np.int32)
    out_c, out_v = [], []
    last = (-1, -1)
    for k in order:
        key = (rows[k], cols[k])
        if key == last:
            out_v[-1] += vals[k]
        else:
            out_c.append(cols[k])
            out_v.append(vals[k])
            rowptr[rows[k] + 1] += 1
            last = key
    rowptr = np.cumsum(rowptr, dtype=np.int64).astype(np.int32)
    return rowptr, np.array(out_c, dtype=np.int32), np.array(out_v, dtype=np.float64)


def csr_to_csc(n_rows, n_cols, rowptr, col, val):
    """Same matrix in the reference's column-major storage (multigrid.h:8-9)."""
    nnz = len(col)
    rows = np.repeat(np.arange(n_rows, dtype=np.int32), np.diff(rowptr))
    order = np.lexsort((rows, col))
    colptr = np.zeros(n_cols + 1, dtype=np.int32)
    np.add.at(colptr, np.asarray(col) + 1, 1)
    colptr = np.cumsum(colptr, dtype=np.int64).astype(np.int32)
    assert colptr[-1] == nnz
    return colptr, rows[order].astype(np.int32), np.asarray(val)[order]


# --------------------------------------------------------------------------
# Grid (grid.h / grid.cpp) -- setup methods only
# --------------------------------------------------------------------------
@dataclass
class GridProperties:  # gridclasses.hpp:6-14
    rbfExp: int = 3
    polyDeg: int = 3
    stencilSize: int = 25
    omega: float = 1.4
    iters: int = 5


@dataclass
class Boundary:  # gridclasses.hpp:15-20
    type: int = 0
    bcPoints: list = field(default_factory=list)
    values: list = field(default_factory=list)


def stencil_size(poly_deg):
    """grid.cpp:266-267 / testing_functions.cpp:378-379."""
    return int(2.5 * (poly_deg + 1) * (poly_deg + 2) / 2)


class Grid:
    def __init__(self, points, boundaries, props, source):
        """grid.cpp:5-27."""
        self.points = np.array(points, dtype=np.float64).reshape(-1, 3)
        self.boundaries = boundaries
        self.props = props
        self.source = np.array(source, dtype=np.float64)
        n = len(self.points)
        self.neumann = any(b.type == 2 for b in boundaries)  # setNeumannFlag :52-60
        self.n = n  # laplaceMatSize_
        self.a_size = n + 1 if self.neumann else n
        self.bcflags = np.zeros(n, dtype=np.int32)
        self.normals = np.zeros((n, 3))
        self.values = np.zeros(self.a_size)
        self.diags = np.zeros(self.a_size)
        self.implicit = False
        self.deriv_normal = []
        self.csr = None
        self.bnd_csr = None

    # grid.cpp:33-40
    def set_bc_flag(self, bnum, kind, values):
        b = self.boundaries[bnum]
        b.type = 1 if kind == "dirichlet" else 2
        for p in b.bcPoints:
            self.bcflags[p] = b.type
        b.values = list(values)

    # grid.cpp:213-260
    def k_nearest(self, ref, neumann, point_bc_flag, k):
        """k smallest (distance, index) pairs; for a boundary point of a Neumann
        grid every *other* boundary point is excluded (grid.cpp:236,244); a point
        at distance exactly 0 is always taken (samePoint)."""
        d = np.sqrt((self.points[:, 0] - ref[0]) ** 2 + (self.points[:, 1] - ref[1]) ** 2)
        allowed = np.ones(self.n, dtype=bool)
        if point_bc_flag and neumann:
            allowed = self.bcflags == 0
        same = np.nonzero(d == 0.0)[0]
        if len(same):
            allowed = allowed.copy()
            allowed[same[-1]] = True
        idx = np.nonzero(allowed)[0]
        order = np.lexsort((idx, d[idx]))[:k]
        return [int(i) for i in idx[order]]

    # grid.cpp:263-303
    def build_coeff_matrix(self, point, neumann, point_bc_flag, poly_deg):
        poly_terms = (poly_deg + 1) * (poly_deg + 2) // 2
        ss = stencil_size(poly_deg)
        nb = self.k_nearest(point, neumann, point_bc_flag, ss)
        sp = shifting_scaling(self.points[nb], point)
        m = np.zeros((ss + poly_terms, ss + poly_terms))
        dx = sp[:ss, 0][:, None] - sp[:ss, 0][None, :]
        dy = sp[:ss, 1][:, None] - sp[:ss, 1][None, :]
        m[:ss, :ss] = np.sqrt(dx * dx + dy * dy) ** self.props.rbfExp
        c = ss
        for p in range(poly_deg + 1):
            for q in range(p + 1):
                pc = sp[:ss, 0] ** (p - q) * sp[:ss, 1] ** q
                m[:ss, c] = pc
                m[c, :ss] = pc
                c += 1
        return m, nb, sp

    # grid.cpp:381-424
    def laplace_weights(self, pid):
        pd, ss = self.props.polyDeg, self.props.stencilSize
        m, nb, sp = self.build_coeff_matrix(self.points[pid], self.neumann, self.bcflags[pid] != 0, pd)
        poly_terms = (pd + 1) * (pd + 2) // 2
        rhs = np.zeros(ss + poly_terms)
        xe, ye = sp[-1, 0], sp[-1, 1]
        mm = float(self.props.rbfExp)
        for i in range(ss):
            xr, yr = sp[i, 0], sp[i, 1]
            d = xe * xe - 2 * xe * xr + xr * xr + ye * ye - 2 * ye * yr + yr * yr
            if d > 0:
                rhs[i] = ((2 * xe - 2 * xr) ** 2 + (2 * ye - 2 * yr) ** 2) * (mm / 2) * (mm / 2 - 1) * d ** (mm / 2 - 2) \
                    + 2 * mm * d ** (mm / 2 - 1)
        r = ss
        for p in range(pd + 1):
            for q in range(p + 1):
                lp = 0.0
                if p - q - 2 >= 0:
                    lp += (p - q) * (p - q - 1) * xe ** (p - q - 2) * ye ** q
                if q - 2 >= 0:
                    lp += q * (q - 1) * xe ** (p - q) * ye ** (q - 2)
                rhs[r] = lp
                r += 1
        w = full_piv_lu_solve(m, rhs)
        scale = sp[-2, 0]
        return w / scale ** 2, nb

    # grid.cpp:304-342 (axis=0) and :343-380 (axis=1)
    def deriv_weights(self, pid, axis):
        pd, ss = self.props.polyDeg, self.props.stencilSize
        m, nb, sp = self.build_coeff_matrix(self.points[pid], self.neumann, self.bcflags[pid] != 0, pd)
        poly_terms = (pd + 1) * (pd + 2) // 2
        rhs = np.zeros(ss + poly_terms)
        xe, ye = sp[-1, 0], sp[-1, 1]
        mm = float(self.props.rbfExp)
        for i in range(1, ss):  # `if (i > 0)` at :320 / :359
            r_ = math.sqrt((sp[i, 0] - xe) ** 2 + (sp[i, 1] - ye) ** 2)
            delta = (xe - sp[i, 0]) if axis == 0 else (ye - sp[i, 1])
            rhs[i] = mm * r_ ** (mm - 2) * delta
        r = ss
        for p in range(pd + 1):
            for q in range(p + 1):
                lp = 0.0
                if axis == 0 and p - q - 1 >= 0:
                    lp += (p - q) * xe ** (p - q - 1) * ye ** q
                if axis == 1 and q - 1 >= 0:
                    lp += q * xe ** (p - q) * ye ** (q - 1)
                rhs[r] = lp
                r += 1
        w = full_piv_lu_solve(m, rhs)
        return w / sp[-2, 0], nb

    # grid.cpp:687-712
    def point_interp_weights(self, point, poly_deg):
        m, nb, sp = self.build_coeff_matrix(point, False, False, poly_deg)
        poly_terms = (poly_deg + 1) * (poly_deg + 2) // 2
        ss = int(2.5 * poly_terms)
        rhs = np.zeros(ss + poly_terms)
        xe, ye = sp[-1, 0], sp[-1, 1]
        rhs[:ss] = np.sqrt((sp[:ss, 0] - xe) ** 2 + (sp[:ss, 1] - ye) ** 2) ** self.props.rbfExp
        r = ss
        for p in range(poly_deg + 1):
            for q in range(p + 1):
                rhs[r] = xe ** (p - q) * ye ** q
                r += 1
        return full_piv_lu_solve(m, rhs), nb

    # grid.cpp:442-461 (unit-square branch only)
    def build_normal_vecs_square(self):
        for p in self.boundaries[0].bcPoints:
            x, y = self.points[p, 0], self.points[p, 1]
            if y == 0:
                self.normals[p] = (0, 1, 0)
            elif y == 1:
                self.normals[p] = (0, -1, 0)
            elif x == 0:
                self.normals[p] = (1, 0, 0)
            elif x == 1:
                self.normals[p] = (-1, 0, 0)

    # grid.cpp:520-548
    def build_deriv_normal_bound(self):
        self.deriv_normal = []
        for b in self.boundaries:
            if b.type != 2:
                continue
            for j, p in enumerate(b.bcPoints):
                wx, nb = self.deriv_weights(p, 0)
                wy, _ = self.deriv_weights(p, 1)
                w = wx * self.normals[p, 0] + self.normals[p, 1] * wy
                self.deriv_normal.append((p, w, nb, b.values[j]))

    # grid.cpp:713-776
    def rcm_order_points(self):
        adj = [self.k_nearest(self.points[i], self.neumann, self.bcflags[i] != 0, self.props.stencilSize)
               for i in range(self.n)]
        if self.neumann and self.implicit:
            for i in range(self.n):
                if self.bcflags[i] != 0:
                    continue
                j = 0
                while j < len(adj[i]):  # list grows while it is scanned (:727-733)
                    a = adj[i][j]
                    if self.bcflags[a] == 2:
                        for k in adj[a]:
                            if k not in adj[i]:
                                adj[i].append(k)
                    j += 1
        order = reverse_cuthill_mckee_ordering(adj)
        self.apply_order(order)
        return order

    def apply_order(self, order):
        """grid.cpp:744-774: new index i holds old point order[i]."""
        order = np.asarray(order, dtype=np.int64)
        assert len(order) == self.n
        old_to_new = np.empty(self.n, dtype=np.int64)
        old_to_new[order] = np.arange(self.n)
        self.points = self.points[order]
        self.bcflags = self.bcflags[order]
        src = self.source.copy()
        src[: self.n] = self.source[order]
        self.source = src
        self.normals = self.normals[order]
        for b in self.boundaries:
            b.bcPoints = [int(old_to_new[p]) for p in b.bcPoints]

    # grid.cpp:549-663
    def build_laplacian(self):
        n = self.n
        trip, btrip = [], []
        for i in range(n):
            w, nb = self.laplace_weights(i)
            if self.bcflags[i] != 2:
                for j, c in enumerate(nb):
                    trip.append((i, c, w[j]))
                    if self.bcflags[i] == 0 and self.bcflags[c] == 2:
                        btrip.append((i, c, w[j]))
                    if i == c:
                        self.diags[i] = w[j]
            if self.neumann and self.bcflags[i] != 2:
                trip.append((i, n, 1.0))
        if self.neumann:
            for i in range(n + 1):
                if i == n or self.bcflags[i] != 2:
                    trip.append((n, i, 1.0))
            for (p, w, nb, _v) in self.deriv_normal:
                for j, c in enumerate(nb):
                    trip.append((p, c, w[j]))
                    if p == c:
                        self.diags[p] = w[j]
        self.csr = csr_from_triplets(self.a_size, self.a_size, trip)
        self.bnd_csr = csr_from_triplets(self.a_size, self.a_size, btrip) if btrip else \
            (np.zeros(self.a_size + 1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
        if not self.implicit:
            return
        rowptr, col, val = self.csr
        rows = len(rowptr) - 1
        for i in range(rows - 1):
            if self.bcflags[i] != 0:
                continue
            bnd = [(int(col[p]), float(val[p])) for p in range(rowptr[i], rowptr[i + 1])
                   if col[p] != rows - 1 and self.bcflags[col[p]] == 2]
            for (jc, a_ij) in bnd:
                a_jj = self.diags[jc]
                for p in range(rowptr[jc], rowptr[jc + 1]):
                    if col[p] == jc:
                        continue
                    trip.append((i, int(col[p]), -val[p] * a_ij / a_jj))
                trip.append((i, jc, -a_ij))
        self.csr = csr_from_triplets(n + 1, n + 1, trip)

    # grid.cpp:62-72
    def modify_coeff_neumann(self, coarse):
        for b in self.boundaries:
            if b.type == 2:
                for j, p in enumerate(b.bcPoints):
                    self.source[p] = 0.0 if coarse else b.values[j]
        self.source[-1] = 0.0

    # grid.cpp:664-685
    def push_inhomog_to_rhs(self):
        if not self.implicit:
            return
        rowptr, col, val = self.bnd_csr
        src = self.source.copy()
        for i in range(self.n):
            if self.bcflags[i] != 0:
                continue
            for p in range(rowptr[i], rowptr[i + 1]):
                self.source[i] -= val[p] * src[col[p]] / self.diags[col[p]]

    # fractionalStepGrid.cpp:60-100 -- D_x, D_y and the plain Laplacian with a row for EVERY point
    def build_fs_matrices(self):
        n = self.n
        out = []
        for kind in ("dx", "dy", "lap"):
            trip = []
            for i in range(n):
                w, nb = (self.deriv_weights(i, 0) if kind == "dx" else
                         self.deriv_weights(i, 1) if kind == "dy" else self.laplace_weights(i))
                for j, c in enumerate(nb):
                    trip.append((i, c, w[j]))
            out.append(csr_from_triplets(n, n, trip))
        return out

    # ---- flattened views for the C oracle / the C-ABI -----------------------
    def boundary_arrays(self):
        btype = np.array([b.type for b in self.boundaries], dtype=np.int32)
        bptr = np.zeros(len(self.boundaries) + 1, dtype=np.int32)
        for i, b in enumerate(self.boundaries):
            bptr[i + 1] = bptr[i] + len(b.bcPoints)
        bpts = np.array([p for b in self.boundaries for p in b.bcPoints], dtype=np.int32)
        bvals = np.array([v for b in self.boundaries for v in b.values], dtype=np.float64)
        return btype, bptr, bpts, bvals


# --------------------------------------------------------------------------
# Multigrid setup (multigrid.cpp:17-60)
# --------------------------------------------------------------------------
def build_interp_matrix(base: Grid, target: Grid, poly_deg: int):
    """multigrid.cpp:17-33 -- (target.n x base.n), returned in CSC like the
    reference's default column-major SparseMatrix, plus the CSR form."""
    trip = []
    for i in range(target.n):
        w, nb = base.point_interp_weights(target.points[i], poly_deg)
        for j, c in enumerate(nb):
            trip.append((i, c, w[j]))
    rowptr, col, val = csr_from_triplets(target.n, base.n, trip)
    colptr, rowidx, cval = csr_to_csc(target.n, base.n, rowptr, col, val)
    return dict(rows=target.n, cols=base.n, colptr=colptr, rowidx=rowidx, val=cval,
                rowptr=rowptr, col=col, rval=val)


def build_matrices(grids, frac_step=False):
    """multigrid.cpp:35-60 (FracStepMultigrid.cpp:17-58 uses the BASE grid's
    polyDeg, :23).  grids sorted coarse -> fine.  Returns (R, P) lists indexed
    like restrictionMatrices_/prolongMatrices_ (R[0] = P[last] = None)."""
    nl = len(grids)
    fine_poly = grids[-1].props.polyDeg
    P = [None] * nl
    R = [None] * nl
    for i in range(nl - 1):
        P[i] = build_interp_matrix(grids[i], grids[i + 1], grids[i].props.polyDeg if frac_step else fine_poly)
    for i in range(1, nl):
        R[i] = build_interp_matrix(grids[i], grids[i - 1], grids[i].props.polyDeg if frac_step else fine_poly)
    for i in range(nl - 1):
        grids[i].modify_coeff_neumann(True)
    return R, P


# --------------------------------------------------------------------------
# synthetic clouds and manufactured problems (testing_functions.cpp)
# --------------------------------------------------------------------------
def square_cloud(nside, seed=12345, jitter=0.25):
    """nside x nside lattice on [0,1]^2, interior points jittered by +-jitter*h,
    boundary coordinates exactly 0 / 1 so that the reference's geometric
    boundary test `x==0||x==1||y==0||y==1` (testing_functions.cpp:86) works."""
    rng = np.random.default_rng(seed)
    h = 1.0 / (nside - 1)
    pts = []
    for j in range(nside):
        for i in range(nside):
            x, y = i * h, j * h
            if i == nside - 1:
                x = 1.0
            if j == nside - 1:
                y = 1.0
            if 0 < i < nside - 1 and 0 < j < nside - 1:
                x += (rng.random() * 2 - 1) * jitter * h
                y += (rng.random() * 2 - 1) * jitter * h
            pts.append((x, y, 0.0))
    return np.array(pts)


def quasi_uniform_square_cloud(nside, offset=0.8):
    """Reference-shaped ("Gmsh-like") cloud of the unit square: the reference reads Gmsh triangulations
    (testing_functions.cpp:355-364) -- evenly spaced boundary nodes (coordinates exactly 0 / 1, :86,180), a
    quasi-uniform interior whose first layer lies about one triangle height off the boundary, a node on every
    corner bisector.  Here: boundary spacing h = 1/(nside-1); hexagonal packing of [offset*h, 1-offset*h]^2 with an
    ODD number of rows so that first and last row are full rows (a point at each of the four inner corners).
    Same point set as the product's `_host.quasi_uniform_square_cloud` (tests/test_host_setup.py)."""
    h = 1.0 / (nside - 1)
    t = [i * h for i in range(nside)]
    t[-1] = 1.0
    pts = [(x, 0.0, 0.0) for x in t] + [(x, 1.0, 0.0) for x in t]
    pts += [(0.0, y, 0.0) for y in t[1:-1]] + [(1.0, y, 0.0) for y in t[1:-1]]
    lo, hi = offset * h, 1.0 - offset * h
    rows = max(3, int(round((hi - lo) / (h * math.sqrt(3.0) / 2.0))) + 1)
    rows += 1 - rows % 2
    m = max(2, int(round((hi - lo) / h)) + 1)
    for r in range(0, rows, 2):
        y = lo + (hi - lo) * r / (rows - 1)
        pts += [(lo + (hi - lo) * k / (m - 1), y, 0.0) for k in range(m)]
    for r in range(1, rows, 2):
        y = lo + (hi - lo) * r / (rows - 1)
        pts += [(lo + (hi - lo) * (k + 0.5) / (m - 1), y, 0.0) for k in range(m - 1)]
    return np.array(pts)


def make_props(poly_deg, omega=1.4, iters=5):
    """testing_functions.cpp:372-380."""
    return GridProperties(rbfExp=3, polyDeg=poly_deg, stencilSize=stencil_size(poly_deg), omega=omega, iters=iters)


def gen_grid_dirichlet_square(points, props, k1=1, k2=1, bvalue_fn=None, order="rcm"):
    """testing_functions.cpp:68-159, geomtype "square".  bvalue_fn lets a test
    prescribe inhomogeneous Dirichlet data (the reference uses 0)."""
    n = len(points)
    src = np.zeros(n)
    bpts, bvals = [], []
    for i, (x, y, _z) in enumerate(points):
        src[i] = -(k1 * k1 + k2 * k2) * PI * PI * math.sin(k1 * PI * x) * math.sin(k2 * PI * y)
        if x == 0 or x == 1 or y == 0 or y == 1:
            bpts.append(i)
            bvals.append(0.0 if bvalue_fn is None else float(bvalue_fn(x, y)))
    g = Grid(points, [Boundary(1, bpts, bvals)], props, src)
    g.implicit = False
    g.set_bc_flag(0, "dirichlet", bvals)
    if order == "rcm":
        g.rcm_order_points()
    g.build_laplacian()
    return g


def gen_grid_neumann_square(points, props, k1=1, k2=1, coarse=False, order="rcm"):
    """testing_functions.cpp:161-284, geomtype "square"."""
    n = len(points)
    src = np.zeros(n + 1)
    bpts, bvals = [], []
    for i, (x, y, _z) in enumerate(points):
        src[i] = -(k1 * k1 + k2 * k2) * PI * PI * math.cos(k1 * PI * x) * math.cos(k2 * PI * y)
        if x == 0 or x == 1 or y == 0 or y == 1:
            bpts.append(i)
            bvals.append(0.0)
    g = Grid(points, [Boundary(2, bpts, bvals)], props, src)
    g.implicit = True
    g.set_bc_flag(0, "neumann", bvals)
    g.build_normal_vecs_square()
    if order == "rcm":
        g.rcm_order_points()
    g.build_deriv_normal_bound()
    g.build_laplacian()
    g.modify_coeff_neumann(coarse)
    g.push_inhomog_to_rhs()
    return g


def calc_l1_error(grid: Grid, x, neumann, k1=1, k2=1):
    """testing_functions.cpp:3-33 (returns the error; mean-shifts a copy)."""
    pts = grid.points
    fn = np.cos if neumann else np.sin
    actual = fn(k1 * PI * pts[:, 0]) * fn(k2 * PI * pts[:, 1])
    v = np.array(x[: grid.n], dtype=np.float64)
    if not neumann:
        return float(np.abs(v - actual).sum() / grid.n)
    v = v + (actual.mean() - v.mean())
    return float(np.abs(v - actual).sum() / grid.n)
