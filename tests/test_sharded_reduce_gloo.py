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
    owned = capi.shard_links(len(w.links), rank, world)
    packed = capi.assemble_packed(K, w.links, CS, _edge_results(orc, w, owned))
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
    owned = [capi.shard_links(len(w.links), r, world) for r in range(world)]
    assert sorted(owned[0] + owned[1]) == list(range(len(w.links))) and not set(owned[0]) & set(owned[1])
    full = capi.assemble_packed(K, w.links, CS, _edge_results(orc, w, range(len(w.links))))
    ref = capi.block_solve(full, K, w.links, B, 1e-3, diag_add=np.full(K * B, 1e-3))
    d0, d1 = (np.load(tmp_path / f"delta_{r}.npy") for r in range(world))
    p0 = np.load(tmp_path / "packed_0.npy")
    assert np.array_equal(d0, d1)                        # every rank solves the identical reduced system
    assert rel(p0, full) < 1e-14 and rel(d0, ref) < 1e-9
    assert np.abs(ref).max() > 0
