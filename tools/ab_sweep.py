#!/usr/bin/env python3
"""Development aid: fine-grid sweep / residual time of ONE library build (MMGP_LIBDIR), for same-box A/B runs
against an older build.  Uses only C-ABI entries that every build has."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshlessmultigridpoisson_amd import _capi, _host  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 216
pts = _host.box_cloud(ns, 3, seed=12345)
tile = _capi.auto_tile_points(ns ** 3, 3, 50, 0, 256, 163840)
g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC, tile_points=tile)
sz = g.sizes()
g.set_source(np.random.default_rng(7).standard_normal(sz["a_size"]))
lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
lv.sweeps(5)
ms = lv.time_sweeps(16, 5)
rs = lv.time_residual(7)
print(json.dumps({"lib": os.environ.get("MMGP_LIBDIR", "tree"), "nside": ns, "tile": tile,
                  "us_per_sweep": float(np.median(ms[1:])) / 16 * 1e3, "resid_us": float(np.median(rs[1:])) * 1e3}))
