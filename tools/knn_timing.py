"""Wall time of mmg_knn (cell-grid build + search + transfers) on a jittered lattice; development aid."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _capi  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = int(sys.argv[2]) if len(sys.argv) > 2 else 216
k = int(sys.argv[3]) if len(sys.argv) > 3 else 50
rng = np.random.default_rng(1)
ax = np.arange(m) / (m - 1.0)
cloud = np.zeros((m ** dim, 3))
cloud[:, :dim] = np.stack(np.meshgrid(*([ax] * dim), indexing="ij"), axis=-1).reshape(-1, dim)
cloud[:, :dim] += 0.25 / (m - 1.0) * rng.uniform(-1, 1, (m ** dim, dim))
_capi.knn(dim, cloud[:1000], cloud[:1000], k)  # warm-up: device init, code object load
for rep in range(2):
    t = time.perf_counter()
    nb = _capi.knn(dim, cloud, cloud, k)
    dt = time.perf_counter() - t
    print(f"mmg_knn dim={dim} n={len(cloud)} k={k}: {dt:.3f} s wall ({len(cloud) / dt / 1e6:.2f} Mqueries/s), self first: {np.mean(nb[:, 0] == np.arange(len(cloud))):.4f}", flush=True)
