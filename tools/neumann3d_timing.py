"""3-D Neumann Poisson hierarchy on the device (edge-free box clouds, scaled multiplier row): residual history and
time per V-cycle; development aid."""
import sys
import time

sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _host as host  # noqa: E402

sides = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "54,108").split(",")]
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ncyc = int(sys.argv[3]) if len(sys.argv) > 3 else 60
theta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
host.set_option("device_setup", 1)
t = time.perf_counter()
clouds = [host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate(sides)]
mg = host.Multigrid(clouds, [deg] * len(sides), dim=3, neumann=True, ordering=host.ORDER_MC, tile_points=0)
print(f"setup {time.perf_counter() - t:.1f} s, points {[len(c) for c in clouds]}", flush=True)
if theta != 1.0:
    mg.set_correction_damping(theta)
res, ms = mg.vcycles(3)
res, ms = mg.vcycles(ncyc)
print(f"{ms / ncyc:.3f} ms per V-cycle; residual history (every 5th): {[f'{r:.2e}' for r in res[::5]]}")
import numpy as np
from meshlessmultigridpoisson_amd import _capi
for l in range(len(sides)):
    g = mg.grid(l)
    sz = g.sizes()
    lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
    info = lv.info()
    rp, col, val = g.csr()
    rl = np.diff(rp)[:sz["n"]]
    ms = lv.time_sweeps(5, 5)
    print("level", l, {k: info[k] for k in ("n_tiles", "n_phases", "lanes_per_row", "waves_per_tile", "max_tile_levels", "sor_rows", "neumann_rows")},
          "row length mean %.1f max %d" % (rl[rl > 0].mean(), rl.max()), "us per sweep %.1f" % (float(np.median(ms[1:])) / 5 * 1e3),
          "us per residual %.1f" % (float(np.median(lv.time_residual(5)[1:])) * 1e3))
