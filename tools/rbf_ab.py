"""A/B of the two stencil-weight kernels (rbf_setup.hip): the register kernel (one wavefront per stencil) against the
LDS kernel (MMG_RBF_KERNEL=lds) on the same stencils -- largest weight difference per shape and kernel time.
The kernel choice is read once per process, so each side runs in its own child process.

    python tools/rbf_ab.py [n_eval]
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

SHAPES = [  # dim, polyDeg, stencil, ops
    (3, 3, 50, [0]),
    (3, 3, 50, [1, 2, 3, 0]),
    (3, 3, 50, [4]),
    (2, 3, 25, [0, 1, 2]),
    (2, 4, 37, [0]),
    (2, 5, 51, [0]),
    (2, 5, 52, [0]),
    (3, 2, 30, [0, 4]),
    (2, 6, 70, [0, 1, 2]),
]


def child(n_eval, out):
    from meshlessmultigridpoisson_amd import _capi
    rng = np.random.default_rng(7)
    res = {}
    for dim, deg, ss, ops in SHAPES:
        side = int(round(n_eval ** (1.0 / dim))) + 1
        ax = [np.arange(side) / (side - 1.0)] * dim
        pts = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, dim)
        pts = pts + (rng.random(pts.shape) - 0.5) * 0.4 / (side - 1.0)
        xyz = np.zeros((pts.shape[0], 3))
        xyz[:, :dim] = pts
        ev = xyz[:n_eval]
        nbr = _capi.knn(dim, xyz, ev, ss)
        _capi.rbf_weights(dim, deg, 3.0, xyz, ev[:64], nbr[:64], ops)  # warm-up
        t0 = time.perf_counter()
        w = _capi.rbf_weights(dim, deg, 3.0, xyz, ev, nbr, ops)
        dt = time.perf_counter() - t0
        key = f"{dim}d_deg{deg}_ss{ss}_ops{len(ops)}"
        res[key] = {"wall_s": dt}
        np.save(os.path.join(out, key + ".npy"), w)
    json.dump(res, open(os.path.join(out, "times.json"), "w"))


def main():
    n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    base = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rbf_ab")  # weights of both sides: too large for gpurun_out
    outs = {}
    for side in ("wave", "one", "lds"):
        out = os.path.join(base, side)
        os.makedirs(out, exist_ok=True)
        env = dict(os.environ, MMG_VERBOSE="1")
        if side != "wave":
            env["MMG_RBF_KERNEL"] = side   # "one": 57..72 unknowns in one wavefront instead of two; "lds": LDS kernel
        p = subprocess.run([sys.executable, __file__, "--child", str(n_eval), out], env=env, capture_output=True, text=True)
        log = [ln for ln in p.stderr.splitlines() if "rbf_weights" in ln]
        open(os.path.join(base, side + ".log"), "w").write(p.stderr)
        if p.returncode:
            print(side, "FAILED", p.returncode, p.stderr[-2000:])
            return 1
        outs[side] = (out, log)
    for dim, deg, ss, ops in SHAPES:
        key = f"{dim}d_deg{deg}_ss{ss}_ops{len(ops)}"
        a = np.load(os.path.join(outs["wave"][0], key + ".npy"))
        b = np.load(os.path.join(outs["lds"][0], key + ".npy"))
        scale = np.abs(b).max(axis=2, keepdims=True)
        err = (np.abs(a - b) / scale).max()
        print(f"{key}: max |wave - lds| / row max = {err:.3e}  finite {np.isfinite(a).all()}")
    for side in ("wave", "one", "lds"):
        print("--", side)
        for ln in outs[side][1]:
            if "stencils of" in ln and " 64 stencils" not in ln:
                print(ln)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), sys.argv[3])
    else:
        sys.exit(main())
