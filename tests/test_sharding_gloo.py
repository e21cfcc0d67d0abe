"""Replica sharding + the final gather for ranking on 2 ranks (gloo, CPU): the N>1 path of
bench.py / the batch driver.  Uses the same code as the GPU run (chromosome3d_amd.sharding)."""
import os
import socket
import sys

import numpy as np
import pytest

from chromosome3d_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_replica_ranges_cover_exactly():
    for total, world in [(20, 1), (20, 2), (20, 4), (20, 8), (7, 3), (3, 8)]:
        got = [sharding.replica_range(total, world, r) for r in range(world)]
        ids = [i for s, c in got for i in range(s, s + c)]
        assert ids == list(range(total))
        assert max(c for _, c in got) - min(c for _, c in got) <= 1
    assert [c for _, c in (sharding.replica_range(20, 8, r) for r in range(8))] == [3, 3, 3, 3, 2, 2, 2, 2]


def test_lpt_assignment_balances_config4():
    # restraint counts of the 22 available 500 kb matrices
    R = [101426, 74211, 69222, 61066, 55278, 47269, 39871, 26525, 34191, 33670, 33153, 17578, 14704, 12874, 11935,
         11628, 10585, 5886, 6670, 2328, 2140, 45150]
    parts = sharding.lpt_assign(R, 8)
    assert sorted(k for p in parts for k in p) == list(range(len(R)))
    loads = [sum(R[k] for k in p) for p in parts]
    assert max(loads) <= max(R) * 1.05 or max(loads) / (sum(R) / 8) < 1.25


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 11
        start, count = sharding.replica_range(5, world, rank)        # 5 replicas over 2 ranks: 3 + 2
        rng = np.random.default_rng(100)
        e_all = rng.uniform(1e4, 2e4, size=5)
        x_all = rng.normal(size=(5, n, 3))
        ids = np.arange(start, start + count)
        rec = sharding.pack_records(ids, e_all[ids], -0.8 - 0.01 * ids, x_all[ids])
        allrec = sharding.gather_records(rec)
        order = sharding.rank_models(allrec)
        q.put((rank, allrec[:, 0].tolist(), allrec[:, 1].tolist(), order, float(np.abs(allrec[:, 4:].reshape(5, n, 3) - x_all).max())))
    finally:
        dist.destroy_process_group()


def test_gather_and_rank_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    rng = np.random.default_rng(100)
    e_all = rng.uniform(1e4, 2e4, size=5)
    expect = sorted(range(5), key=lambda k: (int(e_all[k]), k))
    for rank, ids, e, order, err in res:
        assert ids == [0.0, 1.0, 2.0, 3.0, 4.0]           # every rank sees all replicas, ordered by id
        assert np.allclose(e, e_all) and err == 0.0
        assert order == expect                              # identical ranking on every rank


def _batch_worker(rank, world, port, q):
    """chromosome3d_amd.batch on 2 gloo ranks, the solver replaced by a deterministic stand-in (no GPU here): the
    LPT split, the variable-length all_gather and the per-chromosome ranking are the code the GPU run uses."""
    import torch.distributed as dist
    from chromosome3d_amd import batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sizes = [40, 25, 31, 12, 18]
        costs = [(n - 5) * (n - 4) // 2 for n in sizes]
        mine = sharding.lpt_assign(costs, world)[rank]
        recs = []
        for k in mine:
            rng = np.random.default_rng(1000 + k)                     # what a chromosome yields does not depend on the rank
            r = np.zeros((6, 5))
            r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4] = k, np.arange(6), rng.uniform(1e4, 1.001e4, 6), -rng.uniform(0.7, 0.9, 6), 1.0
            recs.append(r)
        rec = np.concatenate(recs) if recs else np.zeros((0, 5))
        allrec = batch.gather(rec, dist, "cpu")
        per = batch.rank_per_chromosome(allrec, len(sizes))
        q.put((rank, sorted(mine), [[int(v) for v in r[:, 1]] for r in per], [len(r) for r in per]))
    finally:
        dist.destroy_process_group()


def test_chromosome_sharded_batch_world2():
    """Config 4's N > 1 path: every chromosome is solved by exactly one rank, all ranks hold the same per-chromosome
    ranking after the gather, and it equals the single-process ranking."""
    import torch.multiprocessing as mp
    from chromosome3d_amd import batch
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert sorted(res[0][1] + res[1][1]) == [0, 1, 2, 3, 4]            # a partition of the chromosomes
    assert res[0][2] == res[1][2] and res[0][3] == [6] * 5
    expect = []
    for k in range(5):
        rng = np.random.default_rng(1000 + k)
        e = rng.uniform(1e4, 1.001e4, 6)
        expect.append(sorted(range(6), key=lambda r: (int(e[r]), r)))  # ascending int(E_noe), ties by replica id
    assert res[0][2] == expect
    # the restraint-count cost of the bundled 500 kb matrices is what BASELINE.md lists
    st = set()
    m = batch.load_matrices(os.path.join(ROOT, "tests", "golden", "all45"), "_500kb", st)
    # 22 shipped + the chr2_500kb stand-in (N = 479, the largest job): the 23 chromosomes of test.sh:9-12
    assert len(m) == 23 and list(m)[:2] == ["chr1_500kb", "chr2_500kb"] and st == {"chr2_500kb"}
    assert batch.job_costs(m)[0] == 450 * 451 // 2 and batch.job_costs(m)[1] == 474 * 475 // 2
    from chromosome3d_amd import sharding
    assert sharding.lpt_assign(batch.job_costs(m), 8)[0][0] == 1          # the biggest job opens rank 0's list
