#!/usr/bin/env python3
"""bench.py -- fine-grid smoother throughput on MI355X (BASELINE.json metric).

One "step" = one relaxation sweep of the reference's Grid::sor row loop
(grid.cpp:112-145) over the whole fine grid: every interior point updated once,
in the storage (Gauss-Seidel) order, by the hand-written gfx950 sweep kernel.

Workload at N=1: BASELINE.json configs[2] -- 3-D unit cube, 216^3 = 1.008e7 points,
K = 50 neighbours per stencil (3-D polyDeg 3), fp64, Dirichlet, multicolour tile
ordering; the operator is the RBF-FD Laplacian of the reference (Grid::build_laplacian:
PHS r^3 + degree-3 polynomials, one 70 x 70 full-pivot LU per point, batched on the GPU
by mmg_rbf_weights during the untimed setup).  --operator graph selects the synthetic
kNN-graph Laplacian on the same sparsity instead (same bytes per row).  All inputs are
resident in HBM before the timed region.

N>1 (one process per GPU, torch.distributed/RCCL): `--scaling weak` (default): every rank owns
one such 216^3 sub-domain of an N-times larger cloud.  `--scaling strong --total-nside 342`: the N
ranks share ONE 342^3 = 4.0e7-point cloud (BASELINE configs[3] at N = 8), cut into x-slabs.
Launch: `python bench.py --gpus N ...` starts torch.distributed.run itself (one rank per GPU,
127.0.0.1 rendezvous) when it is not already running under it; see DESIGN.md "Multi-GPU".

At N=1 the line also carries two `vcycle` objects: whole Multigrid::vCycle timings (multigrid.cpp:62-110)
on BASELINE configs[1] (2-D 1e6 points, 5 levels) and on the 216^3 4-level hierarchy whose finest grid the
sweep figures are measured on, each with its algorithmic bytes per cycle against 8 TB/s.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def b_sor(k):
    """Algorithmic bytes per relaxed row per sweep: 12*K + 28."""
    return 12 * k + 28


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nside", type=int, default=216, help="points per axis of each rank's cube")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--polydeg", type=int, default=3)
    ap.add_argument("--tile", type=int, default=0, help="points per tile (0 = mmg_auto_tile_points)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per row of the sweep kernel (0 = library default)")
    ap.add_argument("--operator", choices=("rbf", "graph"), default="rbf",
                    help="rbf: the reference's RBF-FD Laplacian (device-batched setup); graph: kNN-graph Laplacian")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--persistent", type=int, default=1,
                    help="0: one launch per phase; 1: automatic (single dependency-driven launch when a sweep needs "
                         "several residency rounds); 4: always single launch; 2: single launch with agent fences")
    ap.add_argument("--force-dd", action="store_true",
                    help="take the domain-decomposition code path (slab-local system, RCCL communicator, exchange "
                         "lists) even with one rank -- rehearsal of the N>1 path on a 1-GPU box")
    ap.add_argument("--slot-bits", type=int, default=12, choices=(12, 16),
                    help="width of the tile-local column indices in the packed matrix stream (A/B)")
    ap.add_argument("--lds-resident", type=int, default=1,
                    help="0: off; 1: LDS-resident tile streams for phases of <= 1 tile per CU; k > 1: up to k tiles per CU (A/B)")
    ap.add_argument("--resid-lds", type=int, default=1, help="0: residual rows stored straight to HBM (A/B)")
    ap.add_argument("--exchange", choices=("sweep", "phase"), default="sweep",
                    help="N>1: ghost refresh once per sweep (block-hybrid Gauss-Seidel, default) or before every phase "
                         "(exact: the sequential reference sweep on the global system)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N>1: weak = --nside^dim points per rank; strong = one --total-nside^dim cloud shared by all ranks")
    ap.add_argument("--total-nside", type=int, default=342, help="--scaling strong: points per axis of the whole cloud")
    ap.add_argument("--partition", choices=("slab", "box"), default="slab",
                    help="N>1: x-slabs (default) or boxes, as cubic as possible (8 ranks: 2 x 2 x 2, SURVEY 8d's decomposition of "
                         "configs[3]; fewer ghost values per rank in strong scaling, up to 7 neighbours instead of 2)")
    ap.add_argument("--no-vcycle", action="store_true", help="skip the whole-V-cycle legs")
    ap.add_argument("--dd-vcycle-nside", type=int, default=128,
                    help="N>1: points per axis of the ONE cloud of the distributed V-cycle leg (every rank builds the global hierarchy)")
    ap.add_argument("--fracstep-nside", type=int, default=72,
                    help="N>1: points per axis of the ONE cloud of the distributed fractional-step leg (configs[4] itself: 216)")
    ap.add_argument("--replicate-below", type=int, default=70000,
                    help="N>1 V-cycle leg: levels with fewer points are kept complete on every rank")
    ap.add_argument("--no-fracstep", action="store_true", help="skip the 3-D fractional-step leg (N=1)")
    ap.add_argument("--dd-legs-timeout", type=float, default=300.0,
                    help="N>1: seconds the optional distributed legs (V-cycle, fractional step) may take together; after that "
                         "rank 0 prints the line with the sweep figures already measured and every rank leaves")
    ap.add_argument("--vcycle-cycles", type=int, default=10)
    ap.add_argument("--verify", type=int, default=0,
                    help="N: run N sweeps in per-phase mode and in --persistent mode from the same state; must agree bitwise")
    return ap.parse_args()


def cpu_baseline(grid, stencil, budget_s, lv=None):
    """The oracle's sweep (plain-C port of grid.cpp:112-145, -O3 -march=native,
    one thread like the reference) on the SAME matrix, for a bounded number of sweeps."""
    from oracle import oracle_c as oc
    try:
        oc.build(fast=True)
        fast = True
    except Exception:
        fast = False
    la = grid.level_arrays()
    olv = oc.Level(la["n"], la["rowptr"], la["col"], la["val"], la["x0"], la["b0"], la["bcflags"], la["neumann"],
                   la["omega"], la["iters"], la["btype"], la["bptr"], la["bpts"], la["bvals"], fast=fast)
    interior = int((la["bcflags"] == 0).sum())
    t0 = time.perf_counter()
    olv.sor_sweeps(1)
    t1 = time.perf_counter() - t0
    n = max(1, min(20, int(budget_s / max(t1, 1e-6)) - 1))
    t0 = time.perf_counter()
    olv.sor_sweeps(n)
    dt = time.perf_counter() - t0
    all_cores = None
    try:  # optional all-cores figure: tiles of one colour concurrently (bitwise the sequential sweep) -- baseline only
        tp = grid.tile_ptr()
        if lv is not None and tp is not None and not la["neumann"]:
            ph = lv.point_phases()
            tile_phase = np.array([max(0, int(ph[tp[t]:tp[t + 1]].max(initial=0))) for t in range(len(tp) - 1)], dtype=np.int32)
            nthreads = usable_cpus()
            lv_par = oc.Level(la["n"], la["rowptr"], la["col"], la["val"], la["x0"], la["b0"], la["bcflags"], la["neumann"],
                              la["omega"], la["iters"], la["btype"], la["bptr"], la["bpts"], la["bvals"], fast=fast)
            lv_par.sor_sweeps_tiled(1, tp, tile_phase, nthreads)
            t0 = time.perf_counter()
            lv_par.sor_sweeps_tiled(8, tp, tile_phase, nthreads)
            all_cores = {"value": interior * 8 / (time.perf_counter() - t0) / 1e6, "unit": "Mpoints/s", "cores": nthreads,
                         "label": "baseline only: colour-parallel sweep over the port's multicolour tiles on POSIX "
                                  "threads (the reference is single-threaded); 8 sweeps"}
    except Exception as e:  # noqa: BLE001
        all_cores = {"error": str(e)}
    return {"value": interior * n / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port", "all_cores": all_cores,
            "sample": f"{n} sweeps over the same {la['n']}-point level (K={stencil}) by the CPU restatement "
                      f"oracle/mmg_oracle.c (the reference itself needs Eigen, absent here: kind 'port'), "
                      f"{'-O3 -march=native -ffp-contract=off' if fast else '-O2'}, 1 thread -- the reference is single-threaded",
            "host": host_description()}


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the container's CPU quota (the GPU box
    shows every core of the node for a share of 16)."""
    from meshlessmultigridpoisson_amd import _capi
    f = _capi.lib().mmg_host_threads
    f.restype = ctypes.c_int
    return max(1, int(f()))


def host_description():
    """nproc, CPU model, compiler: what BASELINE.md section 2 asks to be stated next to the CPU figure."""
    import subprocess
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:  # noqa: BLE001
        cc = "gcc (version unknown)"
    return {"nproc": os.cpu_count(), "usable_cpus": usable_cpus(), "cpu_model": model, "compiler": cc}


def algorithmic_bytes_per_vcycle(levels, k_interp, iters):
    """SURVEY 8d, summed over the levels of one V-cycle (multigrid.cpp:62-110).
    levels: coarse -> fine, dicts with n (points), interior (relaxed rows), K (stencil).
    Per level: 2 * iters sweeps (the coarsest level is smoothed twice as well, :94-95) at 12K + 28 B per
    relaxed row; residual passes at 12K + 24 B per point: 2 on the finest level (:66 and :81), 1 on the
    intermediate ones (:81), none on the coarsest; restriction 12 K_I + 16 B per coarse row, prolongation
    12 K_I + 24 B per fine row, per level pair."""
    total = 0.0
    nl = len(levels)
    for l, lv in enumerate(levels):
        total += 2 * iters * lv["interior"] * (12 * lv["K"] + 28)
        nres = 2 if l == nl - 1 else (1 if l > 0 else 0)
        total += nres * lv["n"] * (12 * lv["K"] + 24)
        if l > 0:
            total += levels[l - 1]["n"] * (12 * k_interp + 16) + lv["n"] * (12 * k_interp + 24)
    return total


def vcycle_leg(mg, what, dim, sides, polys, cycles, iters, oracle_cycles=0):
    """Time device-resident V-cycles of a host Multigrid: 3 warm-up cycles, then 3 batches of `cycles` cycles."""
    from meshlessmultigridpoisson_amd import _capi, _host
    # the cycle body replayed as a HIP graph: one launch per cycle from the host instead of ~60 -- the same kernels, the
    # same bits (tests/test_gpu_configs.py), the same time on a quiet host, but no starvation of the small-level
    # kernels when the host is busy (a 2-D cycle measured 3.3 instead of 2.85 ms on such a box without it)
    _capi.set_option("vcycle_graph", 1)
    try:
        mg.vcycles(3)
        # three batches of `cycles` cycles back to back, the MEDIAN batch is reported (a single 10-cycle window right
        # after the set-up kernels varied 16.6 ... 18.3 ms on the same build; every batch time is in the record)
        batches = []
        res = None
        for _ in range(3):
            rb, ms = mg.vcycles(cycles)
            if res is None:
                res = rb   # contraction: from the FIRST batch (the later ones may already sit on the round-off floor)
            batches.append(ms / cycles)
        ms = float(np.median(batches)) * cycles
    finally:
        _capi.set_option("vcycle_graph", 0)
    levels = []
    for l in range(mg.nlevels):
        g = mg.grid(l)
        sz = g.sizes()
        info = _capi.Level.borrow(g.device_level(), sz["n"], sz["a_size"]).info()
        levels.append({"n": sz["n"], "interior": info["sor_rows"], "K": _host.stencil_size(polys[l], dim),
                       "tiles": info["n_tiles"], "lanes_per_row": info["lanes_per_row"],
                       "waves_per_tile": info["waves_per_tile"]})
    k_i = _host.stencil_size(polys[-1], dim)
    alg = algorithmic_bytes_per_vcycle(levels, k_i, iters)
    per = ms / cycles
    out = {"workload": what, "levels": sides, "polydeg": polys, "cycles": cycles, "ms_per_vcycle": per,
           "ms_per_vcycle_batches": [float(v) for v in batches],
           "algorithmic_bytes_per_vcycle": alg, "achieved": alg / (per * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg / (per * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "fine_Mpoints_per_s": sides[-1] ** dim / (per * 1e-3) / 1e6,
           "layout": [{k: lv[k] for k in ("n", "tiles", "lanes_per_row", "waves_per_tile")} for lv in levels],
           "residual_history_first_cycles": [float(r) for r in mg.residuals[:4]],
           "contraction_per_cycle_timed_cycles": float((res[-1] / res[0]) ** (1.0 / max(1, len(res) - 1))) if len(res) > 1 and res[0] > 0 else None,
           "tolerance_vs_cpu_oracle": "1e-10 relative for rho >= 2e-3, 2e-13 absolute below (evaluation noise of rho); "
                                      "exact-arithmetic mode bitwise (tests/test_gpu_parity.py, tests/test_gpu_configs.py)",
           "sweep_fallbacks": _capi.get_counter("sweep_fallbacks")}
    if oracle_cycles > 0:
        from tests import helpers as H     # cpu_baseline leg: the oracle is the checker / the CPU reference figure
        om = H.oracle_of_multigrid(mg)     # state after the GPU cycles: the continuation is compared
        t0 = time.perf_counter()
        ro = [om.vcycle() for _ in range(oracle_cycles)]
        out["cpu_oracle_ms_per_vcycle"] = (time.perf_counter() - t0) / oracle_cycles * 1e3
        rd = [mg.vcycle() for _ in range(oracle_cycles)]
        out["max_rel_residual_diff_vs_cpu_oracle"] = float(max(abs(x - y) / y for x, y in zip(rd, ro)))
    return out


def fracstep_leg(steps=2, coarse_iters=60):
    """BASELINE configs[4]'s per-GPU share on ONE GPU: the reference's fractional-step time loop
    (FractionalStepSim.cpp:130-156: predictor, PPE source, `while residual >= 1e-10: vCycle; bound_eval_neumann`,
    corrector) device-resident on a 54^3 / 108^3 FractionalStepMultigrid (1.26e6 points = 1e7 / 8), 3-D extension of
    DESIGN section 12 (edge-free clouds, scaled multiplier row); GridProperties::iters of the coarse grid alone is
    raised so that the pressure loop converges in O(150) cycles."""
    from meshlessmultigridpoisson_amd import _host
    t0 = time.perf_counter()
    sides = [54, 108]
    clouds = [_host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate(sides)]
    mg = _host.FracStepMultigrid(clouds, [3, 3], dim=3, dt=1e-3, mu=0.05, rho=1.0, ordering=_host.ORDER_MC, tile_points=0)
    mg.grid(0).set_relaxation(1.4, coarse_iters)
    g = mg.fs_grid()
    g.prescribe_soln()
    g.set_uv_bound()
    t_setup = time.perf_counter() - t0
    from meshlessmultigridpoisson_amd import _capi
    # the ~500 launches of a cycle body (130 sweeps with their boundary solves and multiplier updates) replayed as a
    # HIP graph: +2 % on a quiet host, but a busy host cannot starve the device (a step measured 5.2 instead of 3.5 s on
    # such a box without it)
    _capi.set_option("vcycle_graph", 1)
    try:
        mg.step(max_cycles=3)   # warm-up: device hierarchy, operators
        recs = []
        for _ in range(steps):
            t = time.perf_counter()
            r, nc = mg.step(max_cycles=600)
            recs.append((time.perf_counter() - t, nc, r))
    finally:
        _capi.set_option("vcycle_graph", 0)
    sec = float(np.median([x[0] for x in recs]))
    cyc = int(np.median([x[1] for x in recs]))
    return {"workload": "3-D fractional step (FractionalStepSim.cpp:130-156), 54^3 / 108^3 FractionalStepMultigrid, "
                        f"{g.sizes()['n']} points (BASELINE configs[4]: 1e7 points over 8 GPUs = this per GPU), RBF-FD degree 3 "
                        f"(K=50), dt 1e-3, pressure loop to 1e-10, {coarse_iters} sweeps on the coarse grid, 5 on the fine one",
            "points": g.sizes()["n"], "time_steps": steps, "seconds_per_time_step": sec, "vcycles_per_time_step": cyc,
            "ms_per_vcycle_incl_rest_of_step": sec / max(cyc, 1) * 1e3, "pressure_loop_converged": bool(cyc < 600),
            "fs_residual": float(recs[-1][2]), "setup_seconds": round(t_setup, 1)}


def distributed_vcycle_leg(a, rank, world, dist):
    """Whole V-cycles over `world` GPUs: every rank builds the SAME global hierarchy
    (--dd-vcycle-nside^3 points, 4 levels), keeps its x-slab of every level with at least --replicate-below points and
    a complete copy of the smaller ones (Multigrid::extract_subdomain), registers the exchange lists it worked out
    without communication (Multigrid::setup_exchange) and runs Multigrid::vCycles: ghost refresh before every sweep
    / residual / transfer on the decomposed levels, one ncclAllGather per cycle in front of the first replicated
    level, no collective on the replicated ones.  Time = max over ranks of the device time per cycle."""
    from meshlessmultigridpoisson_amd import _host
    t0 = time.perf_counter()
    ns = a.dd_vcycle_nside
    sides = [max(9, ns // (2 ** (3 - l))) for l in range(4)]
    clouds = [_host.box_cloud(n, 3, seed=777 + i) for i, n in enumerate(sides)]
    _host.set_option("device_setup", 1)
    mg = _host.Multigrid(clouds, [a.polydeg] * 4, dim=3, neumann=False, ordering=_host.ORDER_MC, tile_points=0)
    sub = mg.extract_subdomain(world, rank, replicate_below=a.replicate_below)
    del mg
    sub.setup_exchange_native(exact=False)
    t_setup = time.perf_counter() - t0
    sub.vcycles(2)
    res, ms = sub.vcycles(a.vcycle_cycles)
    per = ms / a.vcycle_cycles
    if dist is not None:
        import torch
        t = torch.tensor([per, t_setup], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per, t_setup = float(t[0].item()), float(t[1].item())
    return {"workload": f"3-D {ns}^3 = {ns ** 3} points in ONE cloud, 4 levels {sides}, RBF-FD degree {a.polydeg}, Dirichlet, omega 1.4, "
                        f"V(5,5), x-slabs over {world} GPUs, levels below {a.replicate_below} points replicated",
            "levels": sides, "replicated": [bool(sub.grid(l).is_replicated()) for l in range(sub.nlevels)],
            "ms_per_vcycle_max_over_ranks": per, "fine_Mpoints_per_s": ns ** 3 / (per * 1e-3) / 1e6,
            "setup_seconds_max_over_ranks": round(t_setup, 1),
            "residual_history": [float(r) for r in res[:4]],
            "contraction_per_cycle": float((res[-1] / res[0]) ** (1.0 / max(1, len(res) - 1))) if res[0] > 0 else None}


def distributed_fracstep_leg(a, rank, world, dist, steps=2, coarse_iters=60):
    """BASELINE configs[4] in form: the fractional-step time loop (FractionalStepSim.cpp:130-156) over `world` GPUs.
    Every rank builds the same global two-level FractionalStepMultigrid (--fracstep-nside^3 points), keeps its x-slab
    as FractionalStepGrid sub-domains (FractionalStepMultigrid::extract_subdomain) and runs mmg_fracstep_step: ghost
    refresh of u, v, w / the hats / the pressure in front of every operator, the distributed pressure loop to 1e-10,
    all-reduced fs_residual.  Time = max over ranks of the wall time per time step."""
    from meshlessmultigridpoisson_amd import _host
    t0 = time.perf_counter()
    ns = a.fracstep_nside
    sides = [ns // 2, ns]
    clouds = [_host.box_cloud(n, 3, seed=12345 + i, edges=False) for i, n in enumerate(sides)]
    mg = _host.FracStepMultigrid(clouds, [3, 3], dim=3, dt=1e-3, mu=0.05, rho=1.0, ordering=_host.ORDER_MC, tile_points=0)
    mg.grid(0).set_relaxation(1.4, coarse_iters)
    g = mg.fs_grid()
    g.prescribe_soln()
    g.set_uv_bound()
    n_glob = g.sizes()["n"]
    sub = mg.extract_subdomain(world, rank)
    del mg
    sub.setup_exchange_native(exact=False)
    t_setup = time.perf_counter() - t0
    sub.step(max_cycles=3)
    recs = []
    for _ in range(steps):
        t = time.perf_counter()
        r, nc = sub.step(max_cycles=600)
        recs.append((time.perf_counter() - t, nc, r))
    sec = float(np.median([x[0] for x in recs]))
    if dist is not None:
        import torch
        t = torch.tensor([sec, t_setup], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec, t_setup = float(t[0].item()), float(t[1].item())
    cyc = int(np.median([x[1] for x in recs]))
    return {"workload": f"3-D fractional step (FractionalStepSim.cpp:130-156) on ONE {ns}^3 cloud ({n_glob} points) over {world} GPUs "
                        f"(x-slabs), levels {sides}, RBF-FD degree 3 (K=50), dt 1e-3, pressure loop to 1e-10, {coarse_iters} sweeps on "
                        f"the coarse grid; BASELINE configs[4] names 1e7 points (--fracstep-nside 216)",
            "points": n_glob, "seconds_per_time_step_max_over_ranks": sec, "vcycles_per_time_step": cyc,
            "pressure_loop_converged": bool(cyc < 600), "fs_residual": float(recs[-1][2]),
            "setup_seconds_max_over_ranks": round(t_setup, 1)}


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torch.distributed.run: start it (one rank per GPU, 127.0.0.1
    rendezvous) as a CHILD process -- nothing in this process has touched the GPU yet -- and relay its one
    JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif r.stdout:
        sys.stderr.write(r.stdout)
    sys.exit(r.returncode if r.returncode else (0 if line else 1))


def main():
    a = parse()
    if (a.gpus > 1 or a.force_dd) and "RANK" not in os.environ:
        spawn_ranks(a)   # does not return (--force-dd: the same path with one rank, rehearsal on a 1-GPU box)
    # stdout carries exactly ONE JSON line.  RCCL prints a version banner on stdout when a communicator is
    # created (torch's and the library's own): everything but the final line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1 or (a.force_dd and "RANK" in os.environ):
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from meshlessmultigridpoisson_amd import _capi, _host
    if world > 1 and "MMG_NUM_THREADS" not in os.environ:
        # one process per GPU: every rank takes its share of the CPUs the job may use (affinity mask capped by the
        # container's quota, not os.cpu_count(): the node shows all its cores) for the (untimed) setup
        os.environ["MMG_NUM_THREADS"] = str(max(2, usable_cpus() // world))
    if _capi.device_count() < 1:
        raise SystemExit("bench.py: no HIP device (libmmgp has no CPU fallback)")
    _capi.check(_capi.lib().mmg_set_device(local_rank))
    _capi.set_option("persistent_sweep", a.persistent)
    _capi.set_option("slot_bits", a.slot_bits)
    _capi.set_option("resid_lds", a.resid_lds)
    _capi.set_option("lds_resident", a.lds_resident)

    # ---- setup (untimed): cloud -> ordering -> operator -> packed device layout ----
    t_setup = time.perf_counter()
    stencil = _host.stencil_size(a.polydeg, a.dim)
    try:
        cus, lds = _capi.device_props()
    except Exception:
        cus, lds = 0, 0
    dd = world > 1 or a.force_dd
    strong = dd and a.scaling == "strong"
    if strong and a.partition == "box":
        pts_per_rank = a.total_nside ** a.dim // max(world, 1)
    elif strong:
        lo_x, hi_x = _host.slab_bounds(rank, world, a.total_nside)
        pts_per_rank = (hi_x - lo_x) * a.total_nside ** (a.dim - 1)
    else:
        pts_per_rank = a.nside ** a.dim
    if a.tile <= 0:
        a.tile = _capi.auto_tile_points(pts_per_rank, a.dim, stencil, a.lanes, cus, lds)
    kind = _host.KIND_DIRICHLET if a.operator == "rbf" else _host.KIND_GRAPH
    if a.operator == "rbf":
        _host.set_option("device_setup", 1)   # the 70 x 70 saddle systems of 1e7 stencils: seconds on the GPU
    mg3 = None
    vcycles = []
    want_vcycle = not dd and not a.no_vcycle and a.dim == 3 and a.operator == "rbf"
    if not dd:
        if want_vcycle:
            # the 4-level hierarchy whose finest grid is the bench level (BASELINE configs[2] + its V-cycle)
            sides3 = [max(9, a.nside // (2 ** (3 - l))) for l in range(4)]
            polys3 = [a.polydeg] * 4
            clouds = [_host.box_cloud(n, 3, seed=12345 + (3 - i)) for i, n in enumerate(sides3)]
            t_fine = time.perf_counter()
            clouds[-1] = _host.box_cloud(a.nside, a.dim, seed=12345)
            t_fine = time.perf_counter() - t_fine
            mg3 = _host.Multigrid(clouds, polys3, dim=3, neumann=False, ordering=_host.ORDER_MC, tile_points=0,
                                  lanes_per_row=a.lanes)
            t_fine += _host.Multigrid.last_setup_times()[3]     # the bench level alone: ordering + operator
            grid = mg3.grid(3)
            a.tile = None
        else:
            pts = _host.box_cloud(a.nside, a.dim, seed=12345)
            grid = _host.Grid.create_square(pts, a.polydeg, dim=a.dim, kind=kind, ordering=_host.ORDER_MC,
                                            tile_points=a.tile, lanes_per_row=a.lanes)
        n_owned = grid.sizes()["n"]
    else:
        # domain decomposition into x-slabs: every rank builds ONLY its own rows from its lattice layers plus a
        # margin; ghost ids are agreed on with one all_gather at setup.  weak: rank r owns layers
        # [r*nside, (r+1)*nside) of a (world*nside) x nside^(dim-1) lattice; strong: an equal share of the
        # layers of ONE total_nside^dim lattice on the unit cube
        local_cloud = _host.block_cloud if a.partition == "box" else _host.slab_cloud
        if strong:
            pts, flags, gid, owner = local_cloud(rank, world, a.total_nside, dim=a.dim, margin=5, total=True)
        else:
            pts, flags, gid, owner = local_cloud(rank, world, a.nside, dim=a.dim, margin=5)
        grid = _host.Grid.create_local(pts, flags, gid, owner, a.dim, stencil, tile_points=a.tile, lanes_per_row=a.lanes,
                                       kind=kind, polydeg=a.polydeg)
        n_owned, lgid, gown = grid.local_map()

        def all_gather_object(obj):
            if dist is None:
                return [obj]
            out = [None] * world
            dist.all_gather_object(out, obj)
            return out

        nbr, sp, si, rp = _host.build_exchange_lists(rank, n_owned, lgid, gown, all_gather_object)
        ids = [_capi.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(ids, src=0)
        _capi.comm_init(rank, world, ids[0])
    sz = grid.sizes()
    t_dev = time.perf_counter()
    lv = _capi.Level.borrow(grid.device_level(), sz["n"], sz["a_size"])
    t_dev = time.perf_counter() - t_dev
    multi = None
    if dd:
        lv.set_exchange(n_owned, nbr, sp, si, rp)
        if a.exchange == "phase":
            lv.set_exchange_mode(1)
    info = lv.info()
    t_setup = time.perf_counter() - t_setup
    if dd:
        # evidence for the N > 1 line: communicator size read back from RCCL, what one ghost refresh moves and costs
        # (HIP events around pack + grouped ncclSend / ncclRecv on the library's stream; collective), setup per rank
        rccl_ranks, rccl_rank = _capi.comm_info()
        n_nbr, n_send, n_recv = lv.exchange_info()
        xms = lv.time_exchange(12)[2:]
        per_rank = all_gather_object({"rank": rank, "rccl_rank": rccl_rank, "setup_seconds": round(t_setup, 1),
                                      "neighbours": n_nbr, "ghost_values_sent": n_send, "ghost_values_received": n_recv,
                                      "exchange_us_median": float(np.median(xms)) * 1e3, "points": int(n_owned)})
        multi = {"rccl_ranks": rccl_ranks, "world_size": world, "per_rank": per_rank,
                 "exchange_us_per_refresh_max_over_ranks": max(r["exchange_us_median"] for r in per_rank),
                 "ghost_bytes_per_refresh_per_rank_max": 8 * max(r["ghost_values_received"] for r in per_rank),
                 "setup_seconds_max_over_ranks": max(r["setup_seconds"] for r in per_rank),
                 "refreshes_per_sweep": 1 if a.exchange == "sweep" else info["n_phases"]}
    # setup_seconds: the bench level (cloud -> ordering -> operator -> packed device layout); the coarser grids and
    # the transfer matrices of the V-cycle leg's hierarchy are reported apart
    t_hier = 0.0
    if mg3 is not None:
        t_hier = t_setup - (t_fine + t_dev)
        t_setup = t_fine + t_dev
    interior = info["sor_rows"]

    # ---- whole V-cycles (N = 1): 216^3 4 levels on the hierarchy just built, then BASELINE configs[1] ----
    if mg3 is not None:
        try:
            vcycles.append(vcycle_leg(mg3, f"3-D {a.nside}^3 = {a.nside ** 3} points, 4 levels {sides3}, RBF-FD degree "
                                           f"{a.polydeg} (K={stencil}) on every level, Dirichlet, omega 1.4, V(5,5) "
                                           f"(the hierarchy above BASELINE configs[2]'s grid)", 3, sides3, polys3,
                                      a.vcycle_cycles, 5, oracle_cycles=0 if a.no_cpu else 1))
        except Exception as e:  # noqa: BLE001 -- the sweep figures below do not depend on this leg
            vcycles.append({"workload": "3-D 4-level V-cycle", "error": str(e)})
    rng = np.random.default_rng(7 + rank)
    rhs = rng.standard_normal(sz["a_size"])
    rhs[n_owned:] = 0.0
    grid.set_source(rhs)
    grid.set_values(np.zeros(sz["a_size"]))
    grid.device_level()   # host writes through values_ / source_ reach the device (Grid::sync_to_device)

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        _capi.check(_capi.lib().mmg_synchronize())

    verify = None
    if a.verify > 0 and a.persistent:
        x0 = lv.get_x()
        _capi.set_option("persistent_sweep", 0)
        lv.sweeps(a.verify)
        xa = lv.get_x()
        lv.set_x(x0)
        _capi.set_option("persistent_sweep", a.persistent)
        lv.sweeps(a.verify)
        xb = lv.get_x()
        verify = {"sweeps": a.verify, "bitwise_equal": bool(np.array_equal(xa, xb)),
                  "max_abs_diff": float(np.abs(xa - xb).max())}
        lv.set_x(x0)

    lv.sweeps(a.warmup)
    barrier()
    t0 = time.perf_counter()
    lv.sweeps(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(interior)], device="cuda", dtype=torch.float64)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_points = float(tot.item())
    else:
        total_points = float(interior)

    # dominant kernel: per-launch HIP events on the library's stream
    sweeps_timed = max(2, min(a.steps, 16))
    kern_ms, launches = lv.time_phases(sweeps_timed)
    alg_bytes = interior * b_sor(stencil) * sweeps_timed
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    # the same kernel as the V-cycle launches it: `iters` = 5 sweeps per smoothing call (multigrid.cpp:79,108), so the
    # ramp-up and the tail of the dependency-driven launch are paid every 5 sweeps instead of every 16
    vc_sweeps = 5
    kern_ms5, launches5 = lv.time_phases(vc_sweeps)
    kern_ms5b, _l5 = lv.time_phases(vc_sweeps)
    kern_ms5 = min(kern_ms5, kern_ms5b)
    achieved5 = interior * b_sor(stencil) * vc_sweeps / (kern_ms5 * 1e-3) / 1e9

    # residual SpMV on the same level (Grid::residual, grid.cpp:147-151, + the two L1 norms of
    # Multigrid::residual): r = b - A x over the same packed stream, HIP events on the library's stream
    res_ms = float(np.median(lv.time_residual(7)[1:]))
    if dist is not None:
        import torch
        t = torch.tensor([res_ms], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res_ms = float(t.item())
    b_res = 12 * stencil + 24
    spmv = {"what": "residual r = b - A x with Dirichlet mask and L1 norms (Grid::residual / Multigrid::residual)",
            "value": total_points / (res_ms * 1e-3) / 1e6, "unit": "Mrows/s", "ms": res_ms,
            "algorithmic_bytes_per_row": b_res,
            "achieved": interior * b_res / (res_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit_bw": "GB/s",
            "frac": interior * b_res / (res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # HBM bytes per launch from the PMC counters of the committed profile of this same workload
    # (profiles/collect.sh: rocprofv3 --pmc passes cannot run inside the timed process)
    traffic, traffic_src = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if (pm["tiles"] == info["n_tiles"] and pm["interior_points_per_gpu"] == interior and pm["stencil"] == stencil
                and launches <= sweeps_timed):
            traffic = pm["bytes_per_launch"] / pm["sweeps_per_launch"] * (sweeps_timed / launches)
            traffic_src = pm["source"]
    except (OSError, KeyError, ValueError):
        pass

    if not dd and not a.no_vcycle and a.operator == "rbf":
        # BASELINE configs[1]: 2-D ~1e6 points, on the reference-shaped (Gmsh-like) cloud, the reference's own parameters
        # (omega 1.4, 5 sweeps, fine polyDeg 4, testing_functions.cpp:372-380).  Three legs:
        #   [0] 5 levels as configs[1] states, sweep point order: contracts slowly (the coarsest grid of 59^2 is only
        #       smoothed, 10 sweeps);  [1] the same cloud coarsened down to ~244 points like the reference's mesh series
        #       (170 -> 600 -> 2.5k -> 10k, x4 per level): 7 levels, contracts 0.6 per cycle;  [2] 5 levels with colour
        #       classes inside the tiles (round 2's order): the fastest schedule, but the cycle DIVERGES at omega 1.4.
        legs2 = [("5 levels as BASELINE configs[1] states", [59, 117, 233, 466, 931], -1),
                 ("7 levels: coarsened like the reference's mesh series down to 244 points", [15, 30, 59, 117, 233, 466, 931], -1),
                 ("5 levels, COLOUR classes inside the tiles (round-2 order; NOT a converging cycle at omega 1.4)",
                  [59, 117, 233, 466, 931], 1)]
        for li, (what, sides2, pcol) in enumerate(legs2):
            try:
                polys2 = [3] * (len(sides2) - 1) + [4]
                t2 = time.perf_counter()
                _host.set_option("point_colouring", pcol)
                try:
                    mg2 = _host.Multigrid([_host.quasi_uniform_square_cloud(n) for n in sides2], polys2, dim=2,
                                          neumann=False, ordering=_host.ORDER_MC, tile_points=0, omega=1.4)
                finally:
                    _host.set_option("point_colouring", -1)
                leg = vcycle_leg(mg2, f"BASELINE configs[1]: 2-D Gmsh-like cloud of {mg2.grid(len(sides2) - 1).sizes()['n']} points "
                                      f"(quasi_uniform_square_cloud({sides2[-1]})), {what}; RBF-FD degree 4 (K=37) on the finest "
                                      f"level, 3 (K=25) below, Dirichlet, omega 1.4, V(5,5) -- the reference's parameters; point "
                                      f"order inside the tiles: {'lexicographic sweep' if pcol != 1 else 'colour classes'}",
                                 2, sides2, polys2, 2 * a.vcycle_cycles, 5, oracle_cycles=0 if (a.no_cpu or li != 0) else 2)
                leg["point_order"] = "sweep" if pcol != 1 else "colour"
                leg["converging"] = bool(leg["contraction_per_cycle_timed_cycles"] is not None and leg["contraction_per_cycle_timed_cycles"] < 1.0)
                leg["setup_seconds"] = round(time.perf_counter() - t2, 1)
                vcycles.insert(li, leg)
                del mg2
            except Exception as e:  # noqa: BLE001
                vcycles.insert(li, {"workload": "BASELINE configs[1]: " + what, "error": str(e)})

    def build_out():
        if strong:
            wl = (f"{a.dim}-D {a.total_nside}^{a.dim} = {a.total_nside ** a.dim} points in ONE cloud shared by {world} "
                  f"GPUs ({'x-slabs' if a.partition == 'slab' else 'boxes'}; BASELINE configs[3] at 8 GPUs)")
        else:
            wl = f"{a.dim}-D {a.nside}^{a.dim} = {n_owned} points per GPU"
        out = {
            "metric": "fine-grid smoother Mpoints/s + achieved HBM GB/s vs roofline, 1/2/4/8 GPU",
            "value": total_points * a.steps / dt / 1e6,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl + f", K={stencil} "
                            f"kNN stencils ({'RBF-FD Laplacian, PHS r^3 + degree-' + str(a.polydeg) + ' polynomials' if a.operator == 'rbf' else 'graph-Laplacian values on the RBF-FD sparsity'}), Dirichlet, "
                            f"one SOR sweep per step" + (" (BASELINE configs[2])" if not strong and a.dim == 3 and a.nside == 216 else ""),
                "points_per_gpu": int(n_owned), "interior_points_per_gpu": int(interior), "stencil": stencil,
                "ordering": "mc_order_points", "tile_points": int(round(n_owned / max(1, info["n_tiles"]))), "tiles": info["n_tiles"],
                "phases_per_sweep": info["n_phases"], "lanes_per_row": info["lanes_per_row"],
                "waves_per_tile": info["waves_per_tile"],
                "lds_bytes_per_wave": info["max_lds_bytes"], "persistent_sweep": int(launches <= sweeps_timed), "sweeps_executed": a.warmup + a.steps + sweeps_timed + 2 * a.verify,
                "packed_bytes_per_row": round((info["stream_bytes"] + 12 * info["halo_entries"]) / max(interior, 1) + 24, 1),
                "parallelism": "single" if not dd else
                               f"domain decomposition: {world} " + ("x-slabs" if a.partition == "slab" else "boxes " + "x".join(str(v) for v in _host.block_dims(world, a.dim)[:a.dim])) + ", RCCL ghost exchange "
                               + ("once per sweep (block-hybrid Gauss-Seidel)" if a.exchange == "sweep" else
                                  "before every phase (exact sequential Gauss-Seidel on the global system)")
                               + f", {sz['n'] - n_owned} ghost values per rank",
                "setup_seconds": round(t_setup, 1),
                "hierarchy_setup_seconds": round(t_hier, 1),
                "sweep_fallbacks": _capi.get_counter("sweep_fallbacks"),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch",
                "traffic_source": traffic_src,
                "kernel": "sweep_persistent_kernel<L,MAXP>" if launches <= sweeps_timed else "tile_kernel<L,MODE_SOR,MAXP>",
                "launches": launches, "sweeps_in_launches": sweeps_timed,
                "avg_launch_us": kern_ms * 1e3 / launches, "us_per_sweep": kern_ms * 1e3 / sweeps_timed,
                "algorithmic_bytes_per_row": b_sor(stencil),
                "frac_in_vcycle": achieved5 / HBM_PEAK_GBS, "achieved_in_vcycle": achieved5,
                "in_vcycle": {"sweeps_per_launch": vc_sweeps, "launches": launches5,
                              "avg_launch_us": kern_ms5 * 1e3 / max(launches5, 1), "us_per_sweep": kern_ms5 * 1e3 / vc_sweeps},
            },
        }
        out["spmv"] = spmv
        if multi is not None:
            out["multi_gpu"] = multi
        if vcycles:
            out["vcycle"] = vcycles
        return out

    # The distributed legs below are optional and full of collectives: should one of them never return (a rank that
    # left a collective the others wait in), the sweep figures above are already measured -- a watchdog lets rank 0
    # print the line without the leg and every rank leave, instead of the whole bench running into the driver's limit.
    import threading
    out_lock = threading.Lock()
    wd = None
    if dd:
        legs_wanted = [k for k, w in (("vcycle", not a.no_vcycle), ("fracstep", not a.no_fracstep))
                       if w and a.operator == "rbf" and a.dim == 3]

        def _legs_expired():
            out_lock.acquire()          # never released: the process ends here
            if rank == 0:
                for k in legs_wanted:
                    if k not in multi:
                        multi[k] = {"error": f"no result within {a.dd_legs_timeout:g} s (bench.py watchdog)"}
                multi["legs_watchdog_fired"] = True
                o = build_out()
                sys.stdout.flush()
                os.dup2(real_stdout, 1)
                print(json.dumps(o), flush=True)
            else:
                time.sleep(2.0)         # rank 0 prints first
            os._exit(0)

        if legs_wanted and a.dd_legs_timeout > 0:
            wd = threading.Timer(a.dd_legs_timeout, _legs_expired)
            wd.daemon = True
            wd.start()
    if dd and not a.no_vcycle and a.operator == "rbf" and a.dim == 3:
        try:
            multi["vcycle"] = distributed_vcycle_leg(a, rank, world, dist)
        except Exception as e:  # noqa: BLE001 -- the sweep figures do not depend on this leg
            multi["vcycle"] = {"error": str(e)}
    if dd and not a.no_fracstep and a.operator == "rbf" and a.dim == 3:
        try:
            multi["fracstep"] = distributed_fracstep_leg(a, rank, world, dist)
        except Exception as e:  # noqa: BLE001
            multi["fracstep"] = {"error": str(e)}

    if wd is not None:
        out_lock.acquire()              # if the watchdog holds it, it is printing: the process ends there
        wd.cancel()
        out_lock.release()

    if rank == 0:
        out = build_out()
        if not dd and not a.no_fracstep and not a.no_vcycle and a.operator == "rbf" and a.dim == 3:
            try:
                out["fracstep"] = fracstep_leg()
            except Exception as e:  # noqa: BLE001 -- the sweep figures do not depend on this leg
                out["fracstep"] = {"workload": "3-D fractional step", "error": str(e)}
        if verify is not None:
            out["config"]["persistent_vs_phase_launches"] = verify
        if not a.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(grid, stencil, a.cpu_seconds, lv)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dd:
        _capi.comm_finalize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
