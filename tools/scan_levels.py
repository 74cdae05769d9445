#!/usr/bin/env python3
"""Development aid: sweep time of one level as a function of (lanes per row, tile points, launch mode).
Not part of the product or of the driver's bench; results go to stdout as JSON lines."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--nsides", type=int, nargs="+", default=[27, 54, 108])
    ap.add_argument("--polydeg", type=int, default=3)
    ap.add_argument("--lanes", type=int, nargs="+", default=[2, 4, 8, 16])
    ap.add_argument("--tiles", type=int, nargs="+", default=[128, 256, 512])
    ap.add_argument("--persistent", type=int, nargs="+", default=[1])
    ap.add_argument("--waves", type=int, nargs="+", default=[1], help="waves per tile (1: packed stream; 2/4/8: dense)")
    ap.add_argument("--sweeps", type=int, default=5)
    ap.add_argument("--opts", type=str, nargs="*", default=[], help="name=value library options")
    a = ap.parse_args()
    from meshlessmultigridpoisson_amd import _capi, _host
    for o in a.opts:
        k, v = o.split("=")
        _capi.set_option(k, int(v))
    K = _host.stencil_size(a.polydeg, a.dim)
    for ns in a.nsides:
        pts = _host.box_cloud(ns, a.dim, seed=12345)
        for T in a.tiles:
            for L, NW in [(l, w) for l in a.lanes for w in a.waves]:
                t0 = time.perf_counter()
                _capi.set_option("waves_per_tile", NW)
                try:
                    g = _host.Grid.create_square(pts, a.polydeg, dim=a.dim, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC,
                                                 tile_points=T, lanes_per_row=L)
                    sz = g.sizes()
                    rng = np.random.default_rng(3)
                    g.set_source(rng.standard_normal(sz["a_size"]))
                    lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
                    info = lv.info()
                except Exception as e:  # noqa: BLE001
                    print(json.dumps({"nside": ns, "T": T, "L": L, "error": str(e)}), flush=True)
                    continue
                for pm in a.persistent:
                    _capi.set_option("persistent_sweep", pm)
                    lv.sweeps(a.sweeps)
                    ms = lv.time_sweeps(a.sweeps, 5)
                    per = float(np.median(ms[1:])) / a.sweeps
                    rows = info["sor_rows"]
                    print(json.dumps({"nside": ns, "n": sz["n"], "T": T, "L": info["lanes_per_row"], "NW": info["waves_per_tile"],
                                      "levels": info["max_tile_levels"], "persistent": pm,
                                      "us_per_sweep": round(per * 1e3, 1),
                                      "frac_hbm": round(rows * (12 * K + 28) / (per * 1e-3) / 8e12, 3),
                                      "tiles": info["n_tiles"], "phases": info["n_phases"], "groups": info["n_groups"],
                                      "groups_per_tile": round(info["n_groups"] / info["n_tiles"], 1),
                                      "stream_B_per_row": round(info["stream_bytes"] / max(rows, 1), 1),
                                      "lds": info["max_lds_bytes"], "setup_s": round(time.perf_counter() - t0, 1)}),
                          flush=True)
                _capi.set_option("persistent_sweep", 1)
                del lv, g


if __name__ == "__main__":
    main()
