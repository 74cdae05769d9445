#!/usr/bin/env python3
"""Development aid (needs the -DMMG_DEBUG_TIMING build: tools/build_dbg.sh, MMGP_LIBDIR=dbglib): timeline of ONE sweep
of a small dense level -- per phase (clusters of the tiles' entry stamps) the median entry, inputs staged, rounds
done, written back; shows what a phase costs besides its rounds.
usage: phase_timeline.py nside neumann(0/1) [dim=3] [polydeg=3]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from meshlessmultigridpoisson_amd import _capi, _host
ns = int(sys.argv[1]); neu = int(sys.argv[2]); dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 3
L = _capi.lib()
L.mmg_debug_timing_tiles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
_host.set_option("device_setup", 1)
pts = _host.box_cloud(ns, dim, seed=12345, edges=not neu) if dim == 3 else _host.quasi_uniform_square_cloud(ns)
g = _host.Grid.create_square(pts, deg, dim=dim, kind=_host.KIND_NEUMANN if neu else _host.KIND_DIRICHLET, ordering=_host.ORDER_MC, tile_points=0)
sz = g.sizes()
lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
info = lv.info()
nt = info["n_tiles"]
lv.sweeps(2)
for rep in range(2):
    ms = lv.time_sweeps(1, 3)
    buf = (C.c_ulonglong * (4 * nt))()
    _capi.check(L.mmg_debug_timing_tiles(buf, nt, int(info["waves_per_tile"] != 1)))
    st = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 4).astype(np.float64) * 0.01
    st = st[st[:, 3] > 0]
    t0 = st[:, 0].min()
    st -= t0
    order = np.argsort(st[:, 0])
    s = st[order]
    cuts = [0] + [i for i in range(1, len(s)) if s[i, 0] - s[i - 1, 0] > 1.5] + [len(s)]
    print(json.dumps({"nside": ns, "neumann": neu, "tiles": nt, "phases": info["n_phases"], "waves": info["waves_per_tile"], "lanes": info["lanes_per_row"],
                      "levels": info["max_tile_levels"], "sweep_us": round(float(ms[-1]) * 1e3, 1), "clusters": len(cuts) - 1}))
    prev_end = 0.0
    for a, b in zip(cuts[:-1], cuts[1:]):
        c = s[a:b]
        print(f"  {b - a:4d} tiles: enter {np.median(c[:, 0]):6.1f} (first {c[:, 0].min():6.1f})  staged +{np.median(c[:, 1] - c[:, 0]):4.1f}  rounds +{np.median(c[:, 2] - c[:, 1]):5.1f} (max {np.max(c[:, 2] - c[:, 1]):5.1f})"
              f"  written +{np.median(c[:, 3] - c[:, 2]):4.1f}  last end {c[:, 3].max():6.1f}  gap from previous last end to first entry {c[:, 0].min() - prev_end:5.1f}")
        prev_end = c[:, 3].max()
