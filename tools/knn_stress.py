"""Randomised stress of mmg_knn against a numpy full scan (bitwise): sizes, k, dimensions, clustered / lattice /
degenerate clouds, flags; development aid (tests/test_gpu_setup.py holds the fixed cases)."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from meshlessmultigridpoisson_amd import _capi
from test_gpu_setup import _brute_knn

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    dim = int(rng.integers(2, 4))
    n = int(rng.integers(1, 4000))
    k = int(rng.integers(1, min(256, max(2, 2 * n)) + 1))
    kind = int(rng.integers(0, 5))
    cloud = np.zeros((n, 3))
    if kind == 0:
        cloud[:, :dim] = rng.uniform(0, 1, (n, dim))
    elif kind == 1:                                   # clusters
        c = rng.uniform(0, 1, (5, dim))
        cloud[:, :dim] = c[rng.integers(0, 5, n)] + 10.0 ** rng.uniform(-6, -1) * rng.standard_normal((n, dim))
    elif kind == 2:                                   # lattice with many ties
        m = max(2, int(round(n ** (1.0 / dim))))
        g = np.stack(np.meshgrid(*([np.arange(m) / m] * dim), indexing="ij"), axis=-1).reshape(-1, dim)
        n = len(g)
        cloud = np.zeros((n, 3))
        cloud[:, :dim] = g[rng.permutation(n)]
        k = min(k, 256)
    elif kind == 3:                                   # collinear / coplanar
        cloud[:, 0] = rng.uniform(0, 1, n)
        if dim == 3:
            cloud[:, 1] = rng.uniform(0, 1e-3, n)
    else:                                             # duplicates
        base = rng.uniform(0, 1, (max(1, n // 3), dim))
        cloud[:, :dim] = base[rng.integers(0, len(base), n)]
    nq = int(rng.integers(1, 60))
    q = np.zeros((nq, 3))
    q[:, :dim] = np.where(rng.uniform(size=(nq, 1)) < 0.5, cloud[rng.integers(0, n, nq), :dim], rng.uniform(-0.5, 1.5, (nq, dim)))
    flags = None
    if rng.uniform() < 0.4:
        flags = ((rng.uniform(size=n) < 0.3).astype(np.uint8), (rng.uniform(size=nq) < 0.5).astype(np.uint8))
    got = _capi.knn(dim, cloud, q, k, *(flags if flags else (None, None)))
    want = _brute_knn(cloud, q, k, dim, *(flags if flags else (None, None)))
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH case", case, dict(dim=dim, n=n, k=k, kind=kind, nq=nq, flags=flags is not None), flush=True)
print("cases done, mismatches:", bad)
