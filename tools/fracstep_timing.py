"""3-D fractional-step time steps on one GPU (BASELINE configs[4]'s per-GPU share: 108^3 = 1.26e6 points); development aid."""
import sys
import time

sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _host as host  # noqa: E402

sides = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "27,54,108").split(",")]
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
host.set_option("device_setup", 1)
import os
from meshlessmultigridpoisson_amd import _capi as _c
_c.set_option("vcycle_graph", int(os.environ.get("GRAPH", "0")))
t = time.perf_counter()
clouds = [host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate(sides)]
mg = host.FracStepMultigrid(clouds, [deg] * len(sides), dim=3, dt=1e-3, mu=0.05, rho=1.0, ordering=host.ORDER_MC, tile_points=0)
if len(sys.argv) > 5:
    mg.grid(0).set_relaxation(1.4, int(sys.argv[5]))   # sweeps on the coarse grid (GridProperties::iters of that grid)
g = mg.fs_grid()
g.prescribe_soln()
g.set_uv_bound()
print(f"setup {time.perf_counter() - t:.1f} s, n = {g.sizes()['n']}", flush=True)
time.sleep(float(os.environ.get('PRE_SLEEP', '0')))
if os.environ.get('WARM'):
    t = time.perf_counter(); print('warm-up step', mg.step(max_cycles=int(os.environ['WARM'])), f'{time.perf_counter() - t:.2f} s', flush=True)
for s in range(steps):
    t = time.perf_counter()
    r, nc = mg.step(max_cycles=int(sys.argv[4]) if len(sys.argv) > 4 else 200)
    dt = time.perf_counter() - t
    from meshlessmultigridpoisson_amd import _capi
    cnt = {k: _capi.get_counter(k) for k in ("plain_cycle_bodies", "graph_launches", "graph_captures", "sweep_fallbacks")}
    print(f"step {s}: fs_residual {r:.6e}, {nc} V-cycles, {dt * 1e3:.1f} ms ({dt * 1e3 / max(nc, 1):.2f} ms per V-cycle incl. the rest of the step) {cnt}", flush=True)
