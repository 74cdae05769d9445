#!/usr/bin/env python3
"""Sweep / residual time of ONE 3-D Dirichlet level (K = 50) under layout options (same-box A/B).
usage: scan_levels3d.py nside [waves ...]   -- waves 0 = automatic, 1 = packed; env DENSE_XTRA=0/1"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshlessmultigridpoisson_amd import _capi, _host
ns = int(sys.argv[1])
_host.set_option("device_setup", 1)
pts = _host.box_cloud(ns, 3, seed=12345)
out = []
for xtra in (2, 0):
    for w in [int(v) for v in sys.argv[2:]] or [0]:
        _capi.set_option("dense_xtra", xtra)
        _capi.set_option("waves_per_tile", w)
        tile = int(os.environ.get("TILE", "0")) or (_capi.auto_tile_points(ns ** 3, 3, 50, 0, 256, 163840) if w != 1 else 0)
        g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_DIRICHLET, ordering=_host.ORDER_MC, tile_points=tile)
        sz = g.sizes()
        lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
        info = lv.info()
        ms = lv.time_sweeps(5, 7)
        us = float(np.median(ms[1:])) / 5 * 1e3
        usr = float(np.median(lv.time_residual(7)[1:])) * 1e3
        rows = info["sor_rows"]
        rec = dict(nside=ns, xtra=xtra, waves_opt=w, waves=info["waves_per_tile"], lanes=info["lanes_per_row"], tiles=info["n_tiles"],
                   stream_B_per_row=round(info["stream_bytes"] / rows, 1), us_per_sweep=round(us, 1), frac=round(rows * 628 / (us * 1e-6) / 8e12, 3),
                   us_resid=round(usr, 1))
        print(json.dumps(rec), flush=True)
        del lv, g
