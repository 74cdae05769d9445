"""Domain decomposition on the CPU: sub-domain extraction, ghost lists, exchange
lists and the block-hybrid Gauss-Seidel schedule (ghosts refreshed once per sweep),
checked against oracle/mmg_oracle.c:orc_sor_hybrid on the GLOBAL system.  The
per-rank arithmetic is done by the CPU interpreter of the packed device plan
(tests/support/plan_emulate.cpp), the exchange by numpy copies (single process) and
by torch.distributed/gloo with world_size 2 (the RCCL path has the same call shape:
one send + one recv per neighbour, ghosts grouped by owner)."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import helpers as H


def _global_problem(host, nside=33, tile=128, seed=21):
    pts = host.square_cloud(nside, seed=seed)
    g = host.Grid.create_square(pts, 3, kind=host.KIND_DIRICHLET, ordering=host.ORDER_MC, tile_points=tile)
    la = g.level_arrays()
    rng = np.random.default_rng(seed)
    la["b0"] = rng.standard_normal(la["a_size"])
    la["x0"] = rng.standard_normal(la["a_size"]) * (la["bcflags"] == 0)
    g.set_source(la["b0"])
    g.set_values(la["x0"])
    return g, la


def _local_level(sub):
    la = sub.level_arrays()
    return H.EmuLevel(la, tile_ptr=sub.tile_ptr(), lanes_per_row=4), la


def _check_against_oracle(la_glob, part, nparts, results, nsweeps):
    o = H.oracle_level(la_glob)
    o.sor_hybrid(part, nparts, nsweeps)
    x = np.array(la_glob["x0"], dtype=np.float64)
    for (n_owned, gid, xloc) in results:
        x[gid[:n_owned]] = xloc[:n_owned]
    assert H.rel_err(x, o.x) < 1e-12
    return o


@pytest.mark.parametrize("nparts", [2, 3])
def test_subdomains_hybrid_schedule_matches_oracle(nparts):
    from meshlessmultigridpoisson_amd import _host as host
    g, la = _global_problem(host)
    part = g.partition_slabs(nparts)
    assert set(np.unique(part)) == set(range(nparts))
    subs = [g.extract_subdomain(part, r) for r in range(nparts)]
    maps = [s.local_map() for s in subs]
    # every global point is owned exactly once; ghosts are foreign points grouped by owner
    owned_all = np.concatenate([gid[:no] for no, gid, _ in maps])
    assert sorted(owned_all.tolist()) == list(range(la["n"]))
    for r, (no, gid, gown) in enumerate(maps):
        assert np.all(part[gid[:no]] == r) and np.all(part[gid[no:]] == gown) and np.all(gown != r)
        assert np.all(np.diff(gown) >= 0)
    needs = []

    def gather(obj):
        needs.append(obj)
        return None

    # two-pass stub of all_gather_object
    for r, (no, gid, gown) in enumerate(maps):
        try:
            host.build_exchange_lists(r, no, gid, gown, gather)
        except TypeError:
            pass
    lists = [host.build_exchange_lists(r, no, gid, gown, lambda obj: needs) for r, (no, gid, gown) in enumerate(maps)]
    levels = [_local_level(s) for s in subs]
    for (e, lal), (no, gid, _) in zip(levels, maps):
        e.x[:] = la["x0"][gid]
        e.b[:] = 0.0
        e.b[:no] = la["b0"][gid[:no]]
    nsweeps = 3
    for _ in range(nsweeps):
        # ghost refresh: values at the end of the previous sweep
        outbox = {}
        for r, (nbr, sp, si, rp) in enumerate(lists):
            for k, q in enumerate(nbr):
                outbox[(r, int(q))] = levels[r][0].x[si[sp[k]:sp[k + 1]]].copy()
        for r, (nbr, sp, si, rp) in enumerate(lists):
            no = maps[r][0]
            for k, q in enumerate(nbr):
                levels[r][0].x[no + rp[k]: no + rp[k + 1]] = outbox[(int(q), r)]
        for e, _ in levels:
            e.sweeps(1)
    _check_against_oracle(la, part, nparts, [(maps[r][0], maps[r][1], levels[r][0].x) for r in range(nparts)], nsweeps)


def test_slab_cloud_local_systems_are_consistent():
    """Weak-scaling path: every rank builds its own part from its owned lattice layers plus
    a margin, without a global system.  Gluing the local matrices together must give one
    consistent global operator, and the hybrid schedule on it must match the oracle."""
    from meshlessmultigridpoisson_amd import _host as host
    nr, nside, dim, K = 2, 14, 3, 50
    subs, maps = [], []
    for r in range(nr):
        pts, flags, gid, owner = host.slab_cloud(r, nr, nside, dim=dim, margin=5)
        s = host.Grid.create_local(pts, flags, gid, owner, dim, K, tile_points=256, lanes_per_row=2)
        subs.append(s)
        maps.append(s.local_map())
    needs = []
    for r, (no, gid, gown) in enumerate(maps):
        needs.append({int(o): gid[no:][gown == o] for o in np.unique(gown)})
    lists = [host.build_exchange_lists(r, no, gid, gown, lambda obj: needs) for r, (no, gid, gown) in enumerate(maps)]
    # glue: global index = offset[r] + local owned index
    offs = np.cumsum([0] + [m[0] for m in maps])
    ntot = int(offs[-1])
    gid2glob = {}
    for r, (no, gid, _) in enumerate(maps):
        for k in range(no):
            gid2glob[int(gid[k])] = int(offs[r] + k)
    assert len(gid2glob) == nr * nside ** 3
    rowptr, col, val, flags_g, part = [0], [], [], np.zeros(ntot, dtype=np.int32), np.zeros(ntot, dtype=np.int32)
    bpts = []
    for r, s in enumerate(subs):
        la = s.level_arrays()
        no, gid, _ = maps[r]
        for k in range(no):
            cs = la["col"][la["rowptr"][k]:la["rowptr"][k + 1]]
            col.extend(gid2glob[int(gid[c])] for c in cs)
            val.extend(la["val"][la["rowptr"][k]:la["rowptr"][k + 1]].tolist())
            rowptr.append(len(col))
        flags_g[offs[r]:offs[r] + no] = la["bcflags"][:no]
        part[offs[r]:offs[r] + no] = r
        bpts.extend((offs[r] + la["bpts"]).tolist())
        assert np.all(la["bcflags"][no:] == 3) and la["rowptr"][no] == la["rowptr"][-1]
    rng = np.random.default_rng(5)
    lag = dict(n=ntot, a_size=ntot, rowptr=np.array(rowptr, dtype=np.int32), col=np.array(col, dtype=np.int32),
               val=np.array(val), bcflags=flags_g, neumann=0, omega=1.4, iters=5, btype=np.array([1], dtype=np.int32),
               bptr=np.array([0, len(bpts)], dtype=np.int32), bpts=np.array(bpts, dtype=np.int32),
               bvals=np.zeros(len(bpts)), x0=rng.standard_normal(ntot) * (flags_g == 0), b0=rng.standard_normal(ntot))
    # every stencil row has K entries and sums to ~0 (graph Laplacian), also across the cut
    A_rows = np.diff(lag["rowptr"])
    assert np.all(A_rows == K)
    levels = []
    for r, s in enumerate(subs):
        e, lal = _local_level(s)
        no, gid, _ = maps[r]
        loc2glob = np.array([gid2glob[int(v)] for v in gid])
        e.x[:] = lag["x0"][loc2glob]
        e.b[:] = 0.0
        e.b[:no] = lag["b0"][loc2glob[:no]]
        levels.append((e, loc2glob))
    for _ in range(2):
        outbox = {}
        for r, (nbr, sp, si, rp) in enumerate(lists):
            for k, q in enumerate(nbr):
                outbox[(r, int(q))] = levels[r][0].x[si[sp[k]:sp[k + 1]]].copy()
        for r, (nbr, sp, si, rp) in enumerate(lists):
            no = maps[r][0]
            for k, q in enumerate(nbr):
                levels[r][0].x[no + rp[k]: no + rp[k + 1]] = outbox[(int(q), r)]
        for e, _ in levels:
            e.sweeps(1)
    o = H.oracle_level(lag)
    o.sor_hybrid(part, nr, 2)
    x = lag["x0"].copy()
    for r, (e, loc2glob) in enumerate(levels):
        no = maps[r][0]
        x[loc2glob[:no]] = e.x[:no]
    assert H.rel_err(x, o.x) < 1e-12


# ---- world_size-2 gloo run ----------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    sys.path.insert(0, H.ROOT)
    from meshlessmultigridpoisson_amd import _host as host
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g, la = _global_problem(host)
    part = g.partition_slabs(world)
    sub = g.extract_subdomain(part, rank)
    no, gid, gown = sub.local_map()

    def all_gather_object(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    nbr, sp, si, rp = host.build_exchange_lists(rank, no, gid, gown, all_gather_object)
    e, _ = _local_level(sub)
    e.x[:] = la["x0"][gid]
    e.b[:] = 0.0
    e.b[:no] = la["b0"][gid[:no]]
    nsweeps = 3
    for _ in range(nsweeps):
        reqs, recv = [], []
        for k, q in enumerate(nbr):
            sb = torch.from_numpy(e.x[si[sp[k]:sp[k + 1]]].copy())
            rb = torch.empty(int(rp[k + 1] - rp[k]), dtype=torch.float64)
            reqs.append(dist.isend(sb, int(q)))
            reqs.append(dist.irecv(rb, int(q)))
            recv.append((k, rb))
        for rq in reqs:
            rq.wait()
        for k, rb in recv:
            e.x[no + rp[k]: no + rp[k + 1]] = rb.numpy()
        e.sweeps(1)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=e.x, gid=gid, n_owned=no, part=part)
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    from meshlessmultigridpoisson_amd import _host as host
    port = _free_port()
    mp.spawn(_gloo_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g, la = _global_problem(host)
    res, part = [], None
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        res.append((int(z["n_owned"]), z["gid"], z["x"]))
        part = z["part"]
    _check_against_oracle(la, part, 2, res, 3)
