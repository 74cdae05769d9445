"""3-D fractional-step: device step vs oracle loop at a size where the oracle still runs; development aid."""
import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import helpers as H
from meshlessmultigridpoisson_amd import _host as host

sides = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "14,27").split(",")]
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ncyc = int(sys.argv[3]) if len(sys.argv) > 3 else 12
host.set_option("device_setup", int(sys.argv[4]) if len(sys.argv) > 4 else 1)
clouds = [host.box_cloud(n, 3, seed=12345 + i) for i, n in enumerate(sides)]
mg = host.FracStepMultigrid(clouds, [deg] * len(sides), dim=3, dt=1e-3, mu=0.05, rho=1.0, ordering=host.ORDER_MC, tile_points=0)
g = mg.fs_grid()
n = g.sizes()["n"]
g.prescribe_soln()
g.set_uv_bound()
om = H.oracle_of_multigrid(mg)
ofs = H.oracle_of_fracstep(g)
comps = [ofs.u, ofs.v, ofs.w]
vecs = [g.vec(0), g.vec(1), g.vec(4)]
for c, vals in zip(comps, vecs):
    c[:] = vals
_bt, _bp, bpts, _bv = g.boundaries()
_xyz, flags = g.points()
arrays = dict(bpts=bpts, bvals=[c[bpts].copy() for c in comps], coupling=g.coupling(), bcflags=flags)
cpu_only = len(sys.argv) > 5 and sys.argv[5] == "cpu"
for step in range(2):
    if cpu_only:
        hist = []
        r0 = om.residual
        r_orc, nc_orc = H.oracle_fracstep_time_step(om, ofs, arrays, g.dt, g.mu, g.rho, 1e-10, ncyc)
        print(f"step {step}: oracle fs_residual {r_orc:.6e} after {nc_orc} cycles; residual now {om.residual():.3e}; |p| max {np.abs(om.levels[-1].x[:n]).max():.3e}", flush=True)
        continue
    r_dev, nc_dev = mg.step(max_cycles=ncyc)
    r_orc, nc_orc = H.oracle_fracstep_time_step(om, ofs, arrays, g.dt, g.mu, g.rho, 1e-10, ncyc)
    print(f"step {step}: device fs_residual {r_dev:.6e} after {nc_dev} cycles; oracle {r_orc:.6e} after {nc_orc}", flush=True)
    print("   device residual history", [f"{v:.3e}" for v in mg.residuals[-nc_dev:]] if mg.residuals else "n/a")
    print("   oracle residual history", [f"{v:.3e}" for v in om.residuals[-nc_orc:]] if hasattr(om, "residuals") else "n/a")
    print("   |p| max device", np.abs(g.values()[:n]).max(), "oracle", np.abs(om.levels[-1].x[:n]).max())
