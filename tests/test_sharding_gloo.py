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


# ---- the EIGHT-rank line, rehearsed ---------------------------------------------------------------------
# Nothing above two ranks had ever executed.  bench.py's main() itself runs here on eight gloo ranks with tests/standin_solver.py in
# the solver's place: the 3,3,3,3,2,2,2,2 split, the max-over-ranks all_reduce of the region clocks, the 8-way variable-length gather
# and the ranking of 20 models, the weak pass, and the config-4 block (23 chromosomes by LPT over 8 ranks, one gather, per-chromosome
# ranking on rank 0).  The real solver runs the same code at four ranks on the GPU box (tests/test_bench_contract.py).
def _bench_rank(rank, world, port, q):
    import io
    from contextlib import redirect_stdout
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      C3D_BENCH_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    import chromosome3d_amd
    from tests.standin_solver import StandinSolver
    chromosome3d_amd.Solver = StandinSolver
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "20", "--warmup", "5", "--reps", "3", "--no-cpu-baseline"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    q.put((rank, buf.getvalue()))


def test_bench_line_at_eight_ranks():
    import json
    import torch.multiprocessing as mp
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_rank, args=(r, 8, port, q)) for r in range(8)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=600) for _ in range(8))
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert all(res[r].strip() == "" for r in range(1, 8))                  # ONE line, from rank 0
    lines = [l for l in res[0].splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 20 and d["warmup"] == 5
    assert d["config"]["replicas_per_gpu"] == [3, 3, 3, 3, 2, 2, 2, 2] and d["models_ranked"] == 20
    assert "20 replicas of chr1_500kb in all: 3+3+3+3+2+2+2+2 per GPU, strong scaling" in d["metric"]
    assert d["collective"] == {"backend": "gloo", "device": "cpu", "world": 8}
    assert d["value"] > 0 and d["weak_scaling_value"] > 0 and "160" not in d["metric"]
    assert d["roofline"]["algorithmic_bytes_per_sa_step"] == 3 * (4 * 101426 + 72 * 455)      # rank 0's GPU: its 3 replicas
    c4 = d["config4"]
    assert c4["chromosomes_ranked"] == 23 and len(c4["per_rank"]) == 8 and sum(r["chromosomes"] for r in c4["per_rank"]) == 23
    assert c4["workload"].startswith("23 chromosomes at 500 kb x 20 replicas (460 models), LPT over 8 rank(s)") and c4["standins"] == ["chr2_500kb"]
    loads = sorted(r["restraints"] for r in c4["per_rank"])
    assert loads[-1] == 474 * 475 // 2 and loads[0] >= 0.6 * loads[-1]       # the largest job (the chr2_500kb stand-in) alone on its rank
    # what rank 0 ranks is what one process would rank: the stand-in's energies are a function of (bead count, replica id) alone
    from tests.standin_solver import StandinSolver
    s = StandinSolver()
    s.n = 455
    s.init_replicas(20, 82364, 0)
    e = s.energies()[:, 0]
    assert abs(d["e_noe_best"] - round(float(e[sorted(range(20), key=lambda k: (int(e[k]), k))[0]]), 1)) < 0.06


def _batch_main_rank(rank, world, port, q):
    import io
    from contextlib import redirect_stdout
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        os.environ.pop(k, None)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), C3D_BENCH_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    from chromosome3d_amd import batch
    from tests.standin_solver import StandinSolver
    batch.Solver = StandinSolver
    buf = io.StringIO()
    with redirect_stdout(buf):
        batch.main(["--json"] + (["--gpus", str(world)] if world > 1 else []))
    q.put((rank, buf.getvalue()))


def test_batch_driver_at_eight_ranks_equals_one_rank():
    """python -m chromosome3d_amd.batch --gpus 8 --json (main() itself, stand-in solver): 23 chromosomes once each, the per-chromosome
    ranking rank 0 prints after the 8-way gather equals the one-process result."""
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    outs = {}
    for world in (1, 8):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=_batch_main_rank, args=(r, world, port, q)) for r in range(world)]
        [p.start() for p in procs]
        res = dict(q.get(timeout=600) for _ in range(world))
        [p.join(120) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        assert all(res[r].strip() == "" for r in range(1, world))
        outs[world] = json.loads([l for l in res[0].splitlines() if l.startswith("{")][-1])
    a, b = outs[1], outs[8]
    assert a["world"] == 1 and b["world"] == 8 and len(a["chromosomes"]) == 23 and a["standins"] == b["standins"] == ["chr2_500kb"]
    assert a["chromosomes"] == b["chromosomes"] and all(len(c["order"]) == 20 for c in b["chromosomes"].values())
