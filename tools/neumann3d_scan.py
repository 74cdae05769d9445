"""Sweep time of a 3-D Neumann level (long rows over several row slots) against tile size and wavefronts per tile;
development aid."""
import sys
import numpy as np
sys.path.insert(0, ".")
from meshlessmultigridpoisson_amd import _capi, _host as host  # noqa: E402
host.set_option("device_setup", 1)
for ns in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "54,108").split(",")]:
    pts = host.box_cloud(ns, 3, seed=12345, edges=False)
    for T in (128, 192, 256, 384, 512, 768):
        for NW in (4, 6):
            _capi.set_option("waves_per_tile", NW)
            try:
                mg = host.Multigrid([pts], [3], dim=3, neumann=True, ordering=host.ORDER_MC, tile_points=T)
                g = mg.grid(0)
                sz = g.sizes()
                lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
                info = lv.info()
                ms = lv.time_sweeps(5, 5)
                print(ns, "T", T, "NW", info["waves_per_tile"], "L", info["lanes_per_row"], "levels", info["max_tile_levels"],
                      "tiles", info["n_tiles"], "us per sweep %.1f" % (float(np.median(ms[1:])) / 5 * 1e3), flush=True)
            except Exception as e:  # noqa: BLE001
                print(ns, T, NW, "error", str(e)[:80], flush=True)
_capi.set_option("waves_per_tile", 0)
