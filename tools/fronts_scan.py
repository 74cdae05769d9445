#!/usr/bin/env python3
"""CPU experiment: contraction of the oracle's V-cycle in the product's tile order with several sweep fronts per tile
(Grid::tile_fronts_).  Checker-side script (uses oracle/): not part of the product."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from meshlessmultigridpoisson_amd import _host as host

ap = argparse.ArgumentParser()
ap.add_argument("--sides", default="49,97,193,385")
ap.add_argument("--deg", type=int, default=4)
ap.add_argument("--neumann", type=int, default=0)
ap.add_argument("--fronts", default="1,2,3,4,6,8")
ap.add_argument("--tiles", default="0")
ap.add_argument("--cycles", type=int, default=14)
ap.add_argument("--tile-order", type=int, default=-1)
a = ap.parse_args()
sides = [int(s) for s in a.sides.split(",")]
host.set_option("tile_order", a.tile_order)
for tile in [int(t) for t in a.tiles.split(",")]:
    for F in [int(f) for f in a.fronts.split(",")]:
        host.set_option("tile_fronts", F)
        mg = host.Multigrid([host.quasi_uniform_square_cloud(s) for s in sides], [3] * (len(sides) - 1) + [a.deg],
                            neumann=bool(a.neumann), ordering=host.ORDER_MC, tile_points=tile)
        om = H.oracle_of_multigrid(mg)
        hist = [om.vcycle() for _ in range(a.cycles)]
        k = a.cycles // 2
        c = (hist[-1] / hist[k]) ** (1.0 / (a.cycles - 1 - k)) if hist[k] > 0 else None
        print(json.dumps({"tile": tile, "fronts": F, "contraction": c, "last": hist[-1]}), flush=True)
