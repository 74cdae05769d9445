import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meshlessmultigridpoisson_amd import _capi, _host
ns, T, L, NW, pm = [int(v) for v in sys.argv[1:6]]
_capi.set_option("waves_per_tile", NW)
pts = _host.box_cloud(ns, 3, seed=12345)
g = _host.Grid.create_square(pts, 3, dim=3, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC, tile_points=T, lanes_per_row=L)
sz = g.sizes()
g.set_source(np.random.default_rng(3).standard_normal(sz["a_size"]))
lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
print("info", lv.info(), flush=True)
_capi.set_option("persistent_sweep", pm)
for k in range(3):
    t0 = time.perf_counter()
    lv.sweeps(2)
    x = lv.get_x()
    print("sweeps", k, round(time.perf_counter() - t0, 3), "fallbacks", _capi.get_counter("sweep_fallbacks"), float(np.abs(x).max()), flush=True)
_capi.set_option("persistent_sweep", 0)
lv2x = lv.get_x()
print("done", flush=True)
