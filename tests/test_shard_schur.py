"""Domain-decomposed solve of a link-sharded window (csrc/shard_solve.cpp, sage_shard_*): W ranks simulated in one
process.  Each rank sees only the packed normal equations of ITS link range, eliminates its interior keyframes and
contributes a Schur complement to the separator buffer; the buffers are summed (what the all-reduce does); every rank
solves the separator system and back-substitutes its keyframes.  The merged delta must equal the single-rank damped
solve of the summed system (sage_block_solve) -- world 2, 4 and 8, temporal windows with and without loop closures,
CS 16 (B = 23) and 32 (B = 39), priors owned by exactly one rank."""
import numpy as np
import pytest

from sage_slam_amd import capi
from tests.helpers import rel


def random_window_system(K, links, B, seed):
    """per-link contributions J^T J (J: m x 2B) in the packed layout, link by link (so that any subset can be summed)"""
    rng = np.random.default_rng(seed)
    BB = B * B
    per_link = []
    for (a, b) in links:
        J = rng.normal(size=(3 * B, 2 * B)) * np.concatenate([np.full(6, 30.0), np.ones(B - 7), [5.0]] * 2)
        H = J.T @ J / (3 * B)
        r = rng.normal(size=3 * B)
        gg = J.T @ r / (3 * B)
        per_link.append((H[:B, :B], H[B:, B:], H[:B, B:], gg[:B], gg[B:]))
    return per_link


def packed_of(K, links, B, per_link, owned):
    BB = B * B
    diag = np.zeros((K, B, B)); lnk = np.zeros((len(links), B, B)); g = np.zeros((K, B))
    for l in owned:
        a, b = links[l]
        Haa, Hbb, Hab, ga, gb = per_link[l]
        diag[a] += Haa; diag[b] += Hbb; lnk[l] += Hab; g[a] += ga; g[b] += gb
    tail = np.array([1.0, 2.0, 3.0, 4.0]) * len(owned)
    return np.concatenate([diag.reshape(-1), lnk.reshape(-1), g.reshape(-1), tail])


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("K,CS,loops", [(64, 32, []), (40, 16, [(0, 39), (3, 30)]), (17, 32, [])])
def test_schur_sharded_solve_equals_single_rank(world, K, CS, loops):
    B = 7 + CS
    links = [(j, i) for i in range(K) for j in range(max(0, i - 3), i)] + loops
    per_link = random_window_system(K, links, B, seed=K + world)
    full = packed_of(K, links, B, per_link, range(len(links)))
    rng = np.random.default_rng(1)
    dadd = np.zeros(K * B); gadd = np.zeros(K * B)
    for k in range(K):
        dadd[k * B + 6:k * B + 6 + CS] = 1e-3
        gadd[k * B + 6:k * B + 6 + CS] = 1e-3 * rng.normal(size=CS)
    dadd[:6] += 1e4; dadd[6 + CS] += 1e4
    damp = 1e-3
    ref = capi.block_solve(full[:-4], K, links, B, damp, dadd, gadd)
    plans = [capi.ShardPlan(K, links, B, r, world) for r in range(world)]
    # ownership: every keyframe has exactly one prior owner; interiors are disjoint; separators shared
    owners = [plans[0].owner(k) for k in range(K)]
    assert all(0 <= o < world for o in owners)
    n_int = sum(p.n_interior for p in plans)
    assert n_int + plans[0].n_sep == K and all(p.n_sep == plans[0].n_sep for p in plans)
    if not loops:
        assert plans[0].n_sep <= 5 * (world - 1)                      # ~3 keyframes per range boundary (<= 5 mid-group)
    sep_sum = np.zeros(plans[0].sep_count)
    for r, p in enumerate(plans):
        owned = capi.shard_links(len(links), r, world)
        sep_sum += p.eliminate(packed_of(K, links, B, per_link, owned), damp, dadd, gadd)
    payload_mb = plans[0].sep_count * 8 / 1e6
    delta = np.full(K * B, np.nan)
    for r, p in enumerate(plans):
        d = p.solve(sep_sum)
        for k in range(K):
            if p.is_local(k):
                seg = d[k * B:(k + 1) * B]
                if not np.isnan(delta[k * B]):
                    assert np.array_equal(delta[k * B:(k + 1) * B], seg)   # separators: identical on every rank
                delta[k * B:(k + 1) * B] = seg
    assert not np.isnan(delta).any()
    print(f"world {world} K {K} B {B}: {plans[0].n_sep} separators, payload {payload_mb:.2f} MB vs packed "
          f"{full.size * 8 / 1e6:.2f} MB; delta vs single-rank solve {rel(delta, ref):.2e}")
    assert rel(delta, ref) < 1e-9
    # the error / inlier totals ride along in the payload tail
    assert np.allclose(sep_sum[-8:-4], full[-4:])
    for p in plans:
        p.close()


@pytest.mark.parametrize("K,CS,loops", [(64, 32, []), (40, 16, [(0, 39), (3, 30)]), (30, 32, [(2, 27)])])
@pytest.mark.parametrize("ndomains", [1, 2, 4])
def test_block_solve_domains_equals_block_solve(K, CS, loops, ndomains):
    """sage_block_solve_domains: the same decomposition inside one process (keyframe-range domains on host threads,
    ONE assembled system; the far end of a long-range link is a separator whatever the domains are)."""
    B = 7 + CS
    links = [(j, i) for i in range(K) for j in range(max(0, i - 3), i)] + loops
    per_link = random_window_system(K, links, B, seed=3 * K + ndomains)
    full = packed_of(K, links, B, per_link, range(len(links)))
    rng = np.random.default_rng(2)
    dadd = np.full(K * B, 1e-3); gadd = 1e-3 * rng.normal(size=K * B)
    dadd[:6] += 1e4
    ref = capi.block_solve(full[:-4], K, links, B, 1e-3, dadd, gadd)
    d = capi.block_solve_domains(full, K, links, B, 1e-3, ndomains, dadd, gadd)
    assert rel(d, ref) < 1e-9
