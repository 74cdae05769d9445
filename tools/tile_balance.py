#!/usr/bin/env python3
"""CPU diagnostic: balance of the tiles of ONE level -- groups (rounds x wavefronts) per tile, per phase the largest
against the mean.  Uses the plan interpreter's library (tests/support): development aid, not part of the product.
usage: tile_balance.py nside [neumann=1] [dim=3] [waves=4] [lanes=16]"""
import os, sys, time
import ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from meshlessmultigridpoisson_amd import _host as host

ns = int(sys.argv[1]); neu = int(sys.argv[2]) if len(sys.argv) > 2 else 1; dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
waves = int(sys.argv[4]) if len(sys.argv) > 4 else 4; lanes = int(sys.argv[5]) if len(sys.argv) > 5 else 16
tile = int(os.environ.get("TILE", "0"))
for k, v in [kv.split("=") for kv in os.environ.get("HOSTOPT", "").split(",") if kv]:
    host.set_option(k, int(v))
t = time.perf_counter()
pts = host.box_cloud(ns, dim, seed=12345, edges=not neu) if dim == 3 else host.quasi_uniform_square_cloud(ns)
g = host.Grid.create_square(pts, 3, dim=dim, kind=host.KIND_NEUMANN if neu else host.KIND_DIRICHLET, ordering=host.ORDER_MC, tile_points=tile)
la = g.level_arrays(1.4, 5)
print(f"setup {time.perf_counter() - t:.1f} s, n {la['n']}", flush=True)
e = H.EmuLevel(la, tile_ptr=g.tile_ptr(), lanes_per_row=lanes, tile_phase=g.tile_phase(), waves_per_tile=waves)
info = e.info()
nt = info["n_tiles"]
L = e.L
L.emu_level_tile_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 3
gr = np.zeros(nt, np.int32); rw = np.zeros(nt, np.int32); ph = np.zeros(nt, np.int32)
ip = C.POINTER(C.c_int)
L.emu_level_tile_stats(e.h, gr.ctypes.data_as(ip), rw.ctypes.data_as(ip), ph.ctypes.data_as(ip))
w = max(1, abs(e.waves())) if e.waves() != 0 else 1
print(info, "waves", e.waves(), "dense_long", e.dense_long(), "stream B/row", L.emu_level_stream_bytes(e.h) / rw.sum())
rounds = gr / w
print(f"rows per tile: mean {rw.mean():.0f} min {rw.min()} max {rw.max()};  rounds per tile: mean {rounds.mean():.1f} min {rounds.min():.0f} max {rounds.max():.0f}")
tot_max = 0; tot_mean = 0
for p in range(info["n_phases"]):
    m = ph == p
    print(f"phase {p}: {m.sum()} tiles, rounds mean {rounds[m].mean():.1f} max {rounds[m].max():.0f}")
    tot_max += rounds[m].max(); tot_mean += rounds[m].mean()
print(f"sum over phases: max {tot_max:.0f}, mean {tot_mean:.0f}  (a phase lasts as long as its slowest tile)")
