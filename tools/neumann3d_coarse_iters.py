"""3-D Neumann two-level hierarchy: convergence and time per cycle against the number of sweeps on the coarse grid
(GridProperties::iters of that grid alone); development aid."""
import sys
import time
sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _host as host  # noqa: E402
sides = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "54,108").split(",")]
host.set_option("device_setup", 1)
clouds = [host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate(sides)]
for coarse_iters in (5, 20, 60, 150):
    mg = host.Multigrid(clouds, [3] * len(sides), dim=3, neumann=True, ordering=host.ORDER_MC, tile_points=0)
    mg.grid(0).set_relaxation(1.4, coarse_iters)
    mg.vcycles(2)
    res, ms = mg.vcycles(40)
    rate = (res[-1] / res[10]) ** (1.0 / (len(res) - 1 - 10))
    print(f"coarse iters {coarse_iters}: {ms / 40:.2f} ms per V-cycle, contraction per cycle {rate:.4f}, residual after 42 cycles {res[-1]:.3e}", flush=True)
