"""N>1 path on CPU (gloo, world_size 2): edge sharding -> packed normal equations -> all-reduce -> host solve.

The per-edge values come from the CPU oracle (allowed in tests); everything else is the product's host logic:
the link-ownership rule (`shard_links`, mirror of sage_window_set_shard), the packed block layout
(`assemble_packed`, mirror of the assemble kernel) and `sage_block_solve`.  The sharded result must equal the
single-rank result: the only cross-rank coupling of the hot path is this one sum."""
import os
import socket

import numpy as np
import pytest

from sage_slam_amd import capi, synth
from tests.helpers import oracle_geo, oracle_photo, rel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _edge_results_directed(orc, w, edges_owned):
    """per-edge oracle results of DIRECTED edges 2 * link + direction (the window engine's ownership unit since r05)"""
    res = {}
    for ge in edges_owned:
        l, d = ge // 2, ge % 2
        a, b = w.links[l]
        k0, k1 = (a, b) if d == 0 else (b, a)
        res[(0, l, d)] = oracle_photo(orc, w, k0, k1)
        res[(1, l, d)] = oracle_geo(orc, w, k0, k1)
    return res


def _edge_results(orc, w, links_owned):
    res = {}
    for l in links_owned:
        a, b = w.links[l]
        for d, (k0, k1) in enumerate(((a, b), (b, a))):
            res[(0, l, d)] = oracle_photo(orc, w, k0, k1)
            res[(1, l, d)] = oracle_geo(orc, w, k0, k1)
    return res


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_window(K=4, H=16, W=20, FS=16, CS=16, L=2, seed=3, back_links=2, border=1, erode=2)
    K, CS, B = len(w.keyframes), w.CS, 7 + w.CS
    owned = capi.shard_edges(len(w.links), rank, world)   # directed edges: a link's two directions may sit on two ranks
    packed = capi.assemble_packed(K, w.links, CS, _edge_results_directed(orc, w, owned))
    t = torch.from_numpy(packed.copy())
    dist.all_reduce(t)                                   # the one data-path collective (sum, double)
    delta = capi.block_solve(t.numpy(), K, w.links, B, 1e-3, diag_add=np.full(K * B, 1e-3))
    np.save(os.path.join(out_dir, f"delta_{rank}.npy"), delta)
    np.save(os.path.join(out_dir, f"packed_{rank}.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_equals_single_rank(orc, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w = synth.make_window(K=4, H=16, W=20, FS=16, CS=16, L=2, seed=3, back_links=2, border=1, erode=2)
    K, CS, B = len(w.keyframes), w.CS, 7 + w.CS
    owned = [capi.shard_edges(len(w.links), r, world) for r in range(world)]
    assert sorted(owned[0] + owned[1]) == list(range(2 * len(w.links))) and not set(owned[0]) & set(owned[1])
    assert len(w.links) % 2 == 1 and owned[0][-1] // 2 == owned[1][0] // 2      # (the odd link is split between the ranks)
    full = capi.assemble_packed(K, w.links, CS, _edge_results(orc, w, range(len(w.links))))
    ref = capi.block_solve(full, K, w.links, B, 1e-3, diag_add=np.full(K * B, 1e-3))
    d0, d1 = (np.load(tmp_path / f"delta_{r}.npy") for r in range(world))
    p0 = np.load(tmp_path / "packed_0.npy")
    assert np.array_equal(d0, d1)                        # every rank solves the identical reduced system
    assert rel(p0, full) < 1e-14 and rel(d0, ref) < 1e-9
    assert np.abs(ref).max() > 0


# ---------------------------------------------------------------------------------------------------------------
# domain-decomposed solve (sage_shard_*, csrc/shard_solve.cpp): the all-reduced payload is the separator Schur system,
# every rank back-substitutes its own keyframes; world 2, 4 and 8 over gloo, per-edge values from the oracle
# ---------------------------------------------------------------------------------------------------------------
SCHUR_WINDOW = dict(K=12, H=16, W=20, FS=16, CS=16, L=2, seed=5, back_links=3, border=1, erode=2)


def _schur_priors(w):
    K, CS, B = len(w.keyframes), w.CS, 7 + w.CS
    dadd = np.zeros(K * B); gadd = np.zeros(K * B)
    for k, kf in enumerate(w.keyframes):
        dadd[k * B + 6:k * B + 6 + CS] = 1e-3
        gadd[k * B + 6:k * B + 6 + CS] = 1e-3 * (0 - kf.code.astype(np.float64))
    dadd[:6] += 1e4
    dadd[6 + CS] += 1e4 / float(w.keyframes[0].scale) ** 2
    return dadd, gadd


def _schur_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    orc.set_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_window(**SCHUR_WINDOW)
    K, CS, B = len(w.keyframes), w.CS, 7 + w.CS
    dadd, gadd = _schur_priors(w)
    owned = capi.shard_links(len(w.links), rank, world)
    packed_local = capi.assemble_packed(K, w.links, CS, _edge_results(orc, w, owned))   # NOT reduced
    plan = capi.ShardPlan(K, w.links, B, rank, world)
    sep = plan.eliminate(packed_local, 1e-3, dadd, gadd)
    t = torch.from_numpy(sep.copy())
    dist.all_reduce(t)                                   # the one data-path collective: the separator system
    delta = plan.solve(t.numpy())
    local = np.array([plan.is_local(k) for k in range(K)])
    np.save(os.path.join(out_dir, f"sdelta_{rank}.npy"), delta)
    np.save(os.path.join(out_dir, f"slocal_{rank}.npy"), local)
    np.save(os.path.join(out_dir, f"stail_{rank}.npy"), t.numpy()[-8:])
    if rank == 0:
        np.save(os.path.join(out_dir, "spayload.npy"), np.array([plan.sep_count, packed_local.size, plan.n_sep]))
    plan.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_schur_sharded_solve_over_gloo(orc, tmp_path, world):
    import torch.multiprocessing as mp
    mp.spawn(_schur_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w = synth.make_window(**SCHUR_WINDOW)
    K, CS, B = len(w.keyframes), w.CS, 7 + w.CS
    dadd, gadd = _schur_priors(w)
    full = capi.assemble_packed(K, w.links, CS, _edge_results(orc, w, range(len(w.links))))
    ref = capi.block_solve(full[:-4], K, w.links, B, 1e-3, dadd, gadd)
    merged = np.full(K * B, np.nan)
    for r in range(world):
        d = np.load(tmp_path / f"sdelta_{r}.npy"); loc = np.load(tmp_path / f"slocal_{r}.npy")
        for k in np.nonzero(loc)[0]:
            seg = d[k * B:(k + 1) * B]
            if not np.isnan(merged[k * B]):
                assert np.array_equal(merged[k * B:(k + 1) * B], seg)   # shared keyframes: bit-identical across ranks
            merged[k * B:(k + 1) * B] = seg
    assert not np.isnan(merged).any()                    # every keyframe is solved by some rank
    nsep_doubles, packed_doubles, nsep = np.load(tmp_path / "spayload.npy")
    print(f"world {world}: {int(nsep)} separator keyframes, payload {nsep_doubles * 8 / 1e3:.0f} kB vs packed "
          f"{packed_doubles * 8 / 1e3:.0f} kB, delta vs single-rank {rel(merged, ref):.2e}")
    assert rel(merged, ref) < 1e-9
    assert np.allclose(np.load(tmp_path / "stail_0.npy")[:4], full[-4:], rtol=1e-12)     # error totals ride along
