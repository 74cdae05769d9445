"""Long residual history of the BASELINE configs[1] hierarchy (2-D 1000^2, 5 levels) on the device; development aid."""
import sys
sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _host as host  # noqa: E402
jit = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
ncyc = int(sys.argv[2]) if len(sys.argv) > 2 else 200
omega = float(sys.argv[3]) if len(sys.argv) > 3 else 1.4
theta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
sides = [62, 125, 250, 500, 1000]
host.set_option("device_setup", 1)
clouds = [host.box_cloud(n, 2, seed=12345 + i, jitter=jit) for i, n in enumerate(sides)]
mg = host.Multigrid(clouds, [3, 3, 3, 3, 4], dim=2, neumann=False, ordering=host.ORDER_MC, tile_points=0, omega=omega)
if theta != 1.0:
    mg.set_correction_damping(theta)
res, ms = mg.vcycles(ncyc)
print("jitter", jit, "omega", omega, "theta", theta, f"{ms / ncyc:.3f} ms per cycle; residual before cycle k, every 20th:", [f"{r:.3e}" for r in res[::20]], "last", f"{res[-1]:.3e}")
