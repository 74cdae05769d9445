#!/usr/bin/env python3
"""Development aid (needs the -DMMG_DEBUG_TIMING build: tools/build_dbg.sh, MMGP_LIBDIR=dbglib):
where a SOR tile spends its time -- stamps per tile: entered / inputs staged / groups done / written back."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--nside", type=int, default=108)
    ap.add_argument("--polydeg", type=int, default=3)
    ap.add_argument("--configs", type=str, nargs="+", default=["256:2"], help="tile:lanes[:waves]")
    ap.add_argument("--modes", type=str, nargs="+", default=["p0", "p1"], help="p<persistent>[l<lds_resident>]")
    a = ap.parse_args()
    from meshlessmultigridpoisson_amd import _capi, _host
    L = _capi.lib()
    L.mmg_debug_timing_tiles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
    pts = _host.box_cloud(a.nside, a.dim, seed=12345)
    for cfg in a.configs:
        T, ln = [int(v) for v in cfg.split(":")[:2]]
        _capi.set_option("waves_per_tile", int(cfg.split(":")[2]) if cfg.count(":") > 1 else 1)
        g = _host.Grid.create_square(pts, a.polydeg, dim=a.dim, kind=_host.KIND_GRAPH, ordering=_host.ORDER_MC,
                                     tile_points=T, lanes_per_row=ln)
        sz = g.sizes()
        g.set_source(np.random.default_rng(3).standard_normal(sz["a_size"]))
        lv = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"])
        info = lv.info()
        nt = min(info["n_tiles"], 1 << 16)
        for mode in a.modes:
            pm = int(mode[1])
            lr = int(mode[3]) if len(mode) > 3 else 1
            _capi.set_option("persistent_sweep", pm)
            _capi.set_option("lds_resident", lr)
            lv.sweeps(2)
            ms = lv.time_sweeps(1, 3)
            buf = (C.c_ulonglong * (4 * nt))()
            _capi.check(L.mmg_debug_timing_tiles(buf, nt, int(info["waves_per_tile"] != 1)))
            st = np.frombuffer(buf, dtype=np.uint64).reshape(nt, 4).astype(np.float64) * 0.01  # us (100 MHz)
            ok = st[:, 3] > 0
            st = st[ok]
            t0 = st[:, 0].min()
            d = {"nside": a.nside, "T": T, "L": info["lanes_per_row"], "NW": info["waves_per_tile"], "mode": mode,
                 "levels": info["max_tile_levels"], "sweep_us": round(float(ms[-1]) * 1e3, 1),
                 "tiles": info["n_tiles"], "groups_per_tile": round(info["n_groups"] / info["n_tiles"], 1),
                 "stage_us_med": round(float(np.median(st[:, 1] - st[:, 0])), 2),
                 "groups_us_med": round(float(np.median(st[:, 2] - st[:, 1])), 2),
                 "write_us_med": round(float(np.median(st[:, 3] - st[:, 2])), 2),
                 "tile_us_med": round(float(np.median(st[:, 3] - st[:, 0])), 2),
                 "tile_us_p90": round(float(np.percentile(st[:, 3] - st[:, 0], 90)), 2),
                 "span_us": round(float(st[:, 3].max() - t0), 1)}
            d["per_group_us"] = round(d["groups_us_med"] / d["groups_per_tile"], 3)
            print(json.dumps(d), flush=True)
        _capi.set_option("persistent_sweep", 1)
        _capi.set_option("lds_resident", 1)
        del lv, g


if __name__ == "__main__":
    main()
