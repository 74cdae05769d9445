#!/usr/bin/env python3
"""bench_vcycle.py -- whole V-cycle timing (not the driver's bench line).
BASELINE.json configs[1]: 2-D ~1e6-point cloud, 5-level V-cycle, fp64, real RBF-FD
Laplacians (fine polyDeg 4, coarse 3) and RBF interpolation transfers, built by the
host classes; V-cycles run device-resident through Multigrid::vCycles."""
import argparse
import json
import sys
import time
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nside", type=int, default=1000)
    ap.add_argument("--dim", type=int, default=2, help="3: box clouds (BASELINE configs[2] with --nside 216 --levels 4 --polydeg 3)")
    ap.add_argument("--levels", type=int, default=5)
    ap.add_argument("--cycles", type=int, default=20)
    ap.add_argument("--polydeg", type=int, default=4)
    ap.add_argument("--persistent", type=int, default=1)
    ap.add_argument("--omega", type=float, default=1.4, help="SOR factor (reference default 1.4, testing_functions.cpp)")
    ap.add_argument("--iters", type=int, default=5, help="sweeps per smoothing call (reference default 5)")
    ap.add_argument("--lds-resident", type=int, default=1, help="0: plain per-phase kernel on small levels (A/B)")
    ap.add_argument("--graph", type=int, default=1, help="0: issue every launch of the cycle body directly (A/B of the HIP graph)")
    ap.add_argument("--waves", type=int, default=0, help="waves_per_tile option: 0 automatic, 1 packed stream everywhere (A/B)")
    ap.add_argument("--point-colouring", type=int, default=-1, help="-1 automatic (2-D: 2, 3-D: 1); 0 / 1: point colours inside the tiles; 2: lexicographic SWEEP order inside the tiles")
    ap.add_argument("--dense-xtra", type=int, default=1, help="0: 16 x 4 entries per dense row slot instead of 16 x 3 + 1 (A/B)")
    ap.add_argument("--dense-single", type=int, default=1, help="0: keep multi-wavefront rounds on sweep-ordered levels (A/B)")
    ap.add_argument("--tile-order", type=int, default=-1, help="-1 automatic (2-D Neumann: 1); 0: tile colours; 1: tiles in lexicographic sweep order")
    ap.add_argument("--tile", type=int, default=0, help="points per tile (0: automatic)")
    ap.add_argument("--cloud", default="jitter", choices=["jitter", "gmsh"], help="gmsh: quasi_uniform_square_cloud (2-D only)")
    ap.add_argument("--neumann", type=int, default=0)
    ap.add_argument("--ordering", default="mc", choices=["mc", "rcm"], help="rcm: the reference's rcm_order_points")
    ap.add_argument("--sides", type=str, default="", help="explicit comma-separated sides, coarse to fine (overrides --nside/--levels)")
    ap.add_argument("--per-level", type=str, default="", help="write a per-level table (sweep, residual: us, %% of 8 TB/s) to this markdown file")
    a = ap.parse_args()
    from meshlessmultigridpoisson_amd import _capi, _host
    _capi.set_option("persistent_sweep", a.persistent)
    _capi.set_option("lds_resident", a.lds_resident)
    _capi.set_option("vcycle_graph", a.graph)
    _capi.set_option("waves_per_tile", a.waves)
    _capi.set_option("dense_single", a.dense_single)
    _capi.set_option("dense_xtra", a.dense_xtra)
    _host.set_option("point_colouring", a.point_colouring)
    _host.set_option("tile_order", a.tile_order)
    t0 = time.perf_counter()
    sides = [max(9, a.nside // (2 ** (a.levels - 1 - l))) for l in range(a.levels)]
    if a.sides:
        sides = [int(v) for v in a.sides.split(",")]
        a.levels = len(sides)
    if a.dim == 3:
        clouds = [_host.box_cloud(n, 3, seed=12345 + i, edges=not a.neumann) for i, n in enumerate(sides)]
    elif a.cloud == "gmsh":
        clouds = [_host.quasi_uniform_square_cloud(n) for n in sides]
    else:
        clouds = [_host.square_cloud(n, seed=12345 + i) for i, n in enumerate(sides)]
    polys = [3] * (a.levels - 1) + [a.polydeg]
    mg = _host.Multigrid(clouds, polys, dim=a.dim, neumann=bool(a.neumann),
                         ordering=_host.ORDER_MC if a.ordering == "mc" else _host.ORDER_RCM, tile_points=a.tile,
                         omega=a.omega, iters=a.iters)
    t_setup = time.perf_counter() - t0
    res, ms = mg.vcycles(3)  # warm-up, creates the device hierarchy
    t0 = time.perf_counter()
    res, ms = mg.vcycles(a.cycles)
    wall = time.perf_counter() - t0
    n_fine = mg.grid(mg.nlevels - 1).sizes()["n"]
    out = {"workload": f"{a.dim}-D {a.cloud} cloud, {n_fine} points, {a.levels} levels {sides}, polyDeg {polys}, omega {a.omega}, iters {a.iters}, "
                       f"{'Neumann' if a.neumann else 'Dirichlet'}, ordering {a.ordering} (point order {a.point_colouring}, tile order {a.tile_order}, tile {a.tile})",
           "contraction_per_cycle": float((res[-1] / res[len(res) // 2]) ** (1.0 / max(1, len(res) - 1 - len(res) // 2))) if res[len(res) // 2] > 0 else None,
           "setup_seconds": round(t_setup, 1), "cycles": a.cycles, "device_ms_per_vcycle": ms / a.cycles,
           "wall_ms_per_vcycle": wall / a.cycles * 1e3, "residuals": [float(r) for r in mg.residuals[:8]], "residual_before_last_cycle": float(res[-1]),
           "fine_points_per_s_per_vcycle": n_fine / (ms / a.cycles * 1e-3)}
    if a.per_level:
        import numpy as np
        lines = [f"## {out['workload']}", "",
                 f"device {out['device_ms_per_vcycle']:.3f} ms per V-cycle (V({a.iters},{a.iters}), {a.cycles} cycles back to back)", "",
                 "| level | points | relaxed rows | K | layout (tile points x lanes per row x wavefronts per tile) | tiles | phases | "
                 "us per sweep | % of 8 TB/s (sweep) | us per residual | % of 8 TB/s (residual) | share of the cycle: 2 x iters sweeps + residual |",
                 "|---|---|---|---|---|---|---|---|---|---|---|---|"]
        for l in range(mg.nlevels):
            g = mg.grid(l)
            sz = g.sizes()
            lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
            info = lv.info()
            K = _host.stencil_size(polys[l], a.dim)
            rows = info["sor_rows"]
            ms_s = lv.time_sweeps(a.iters, 7)
            us_sweep = float(np.median(ms_s[1:])) / a.iters * 1e3
            ms_r = lv.time_residual(7)
            us_res = float(np.median(ms_r[1:])) * 1e3
            alg = rows * (12 * K + 28)
            share = (2 * a.iters * us_sweep + us_res) / (out["device_ms_per_vcycle"] * 1e3)
            lines.append(f"| {l} | {sz['n']} | {rows} | {K} | {sz['n'] // max(1, info['n_tiles'])} x {info['lanes_per_row']} x {info['waves_per_tile']} | "
                         f"{info['n_tiles']} | {info['n_phases']} | {us_sweep:.1f} | {alg / (us_sweep * 1e-6) / 8e12 * 100:.1f} | "
                         f"{us_res:.1f} | {rows * (12 * K + 24) / (us_res * 1e-6) / 8e12 * 100:.1f} | {share * 100:.1f} % |")
        with open(a.per_level, "a") as f:
            f.write("\n".join(lines) + "\n\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
