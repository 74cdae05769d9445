import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from meshlessmultigridpoisson_amd import _capi
from test_gpu_setup import _brute_knn
dim, n, k, jitter = 2, 3000, 28, 0.0
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
print("unflagged equal", np.array_equal(got, want))
cflag = (rng.uniform(size=len(cloud)) < 0.3).astype(np.uint8)
qflag = np.concatenate([cflag[sel], np.ones(20, np.uint8), np.zeros(20, np.uint8)])
got = _capi.knn(dim, cloud, queries, k, cflag, qflag)
want = _brute_knn(cloud, queries, k, dim, cflag, qflag)
bad = np.flatnonzero((got != want).any(axis=1))
print("bad rows", len(bad), bad[:20])
for e in bad[:4]:
    q = queries[e]
    print("query", e, q, "qflag", qflag[e])
    dg = np.sqrt(((cloud[got[e]] - q) ** 2).sum(axis=1))
    dw = np.sqrt(((cloud[want[e]] - q) ** 2).sum(axis=1))
    for j in range(k):
        mark = "" if got[e, j] == want[e, j] else "   <--"
        print(f"  {j:3d} got {got[e, j]:5d} d={dg[j]:.17g} f={cflag[got[e, j]]}   want {want[e, j]:5d} d={dw[j]:.17g} f={cflag[want[e, j]]}{mark}")
