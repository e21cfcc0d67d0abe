"""Chromosome-sharded batch (BASELINE.json configs[3]: every 500 kb chromosome x 20 replicas over the GPUs of a node).

    python -m chromosome3d_amd.batch [--gpus N] [--inputs tests/golden/all45] [--pattern _500kb] [--models 20] [--out DIR]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m chromosome3d_amd.batch --gpus 8 ...
(`--gpus N` alone starts the N ranks itself, as child processes, before any HIP call: launch.py)

One process per GPU (the reference's only concurrency is one process per chromosome, test.sh:4-12).  Chromosomes go
to ranks by longest-processing-time-first on their restraint counts (sharding.lpt_assign); every rank solves its
chromosomes through the C ABI with `models` replicas each (replica ids 0..models-1, Philox keyed by (seed, replica):
the models of a chromosome do not depend on which rank solved it); one all_gather of the per-model records
{chromosome, replica, E_noe, Spearman, anneal ms} (RCCL on GPUs; C3D_BENCH_BACKEND=gloo for a rehearsal with fewer
GPUs than ranks) brings everything to rank 0, which ranks the models of every chromosome as chromosome3D.pl:796-802
does.  With --out every rank also writes the reference's output files for its chromosomes (pipeline.assess_dgsa).

The single-process, thread-per-GPU twin of this module is the native c3d_batch executable (csrc/c3d_batch_main.cpp).
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import numpy as np

from . import pipeline, sharding
from .solver import Solver, default_model, default_schedule


def load_matrices(inputs, pattern, standins=None):
    """{chromosome id: IF matrix} from a directory of packed upper triangles (*_upper.npz, tests/golden/all45) or of
    the reference's text matrices (*_matrix.txt), filtered by `pattern`, in chromosome order.  `standins` (a set) receives
    the ids of packed matrices marked `standin` — chr2_500kb, which the reference does not ship (.MISSING_LARGE_BLOBS:1;
    tools/make_chr2_standin.py): part of the workload (test.sh:9-12 runs 23 chromosomes), never of a parity table."""
    mats = {}
    for p in glob.glob(os.path.join(inputs, "*_upper.npz")):
        cid = os.path.basename(p)[:-len("_upper.npz")]
        if pattern in cid:
            z = np.load(p)
            if standins is not None and "standin" in z.files:
                standins.add(cid)
            n = int(z["n"])
            m = np.zeros((n, n))
            iu = np.triu_indices(n)
            m[iu] = z["upper"]
            m.T[iu] = z["upper"]
            mats[cid] = m
    for p in glob.glob(os.path.join(inputs, "*_matrix.txt")):
        cid = os.path.basename(p)[:-len("_matrix.txt")]
        if pattern in cid and cid not in mats:
            mats[cid] = pipeline.parse_if_file(p)

    def key(c):
        m = re.match(r"chr(\d+)_(\w+)", c)
        return (m.group(2), int(m.group(1))) if m else (c, 0)
    return {c: mats[c] for c in sorted(mats, key=key)}


def job_costs(mats):
    """~ restraint count of every chromosome: pairs i < j with |i - j| >= 5."""
    return [(m.shape[0] - 5) * (m.shape[0] - 4) // 2 for m in mats.values()]


PAIR_MAX_BEADS = 270      # a four-XCD geometry exists up to 288 beads (5 replicas per XCD x 6 workgroups of 48 rows); it pays up to ~270 (profiles/r05_config4_paired_anneals.txt: N = 287 anneals in 16.3 ms on half a device, 13.2 on all of it)


def _prepare(solver, IF, models, seed, min_steps, gtol, xcds):
    solver.set_option("cluster_xcd_base", 0)
    solver.set_option("cluster_xcd_count", xcds)
    solver.set_model(default_model())
    d10 = pipeline.IF2dist_new(solver, IF)
    solver.set_schedule(default_schedule(min_steps), None, gtol, 250)
    solver.init_replicas(models, seed, 0)
    return d10


def solve_assigned(solver, mats, mine, models=20, seed=82364, min_steps=3000, gtol=1e-2, out=None, on_job=None, second=None):
    """Solve the chromosomes with indices `mine`; returns records [len(mine) * models, 5]:
    chromosome index, replica, E_noe, Spearman(IF, d), anneal ms.  on_job(cid, solver) is called after every anneal (bookkeeping hooks).
    `second` = another context on the same GPU: two consecutive chromosomes of at most PAIR_MAX_BEADS beads then anneal SIDE BY SIDE, one on
    XCDs 0-3, the other on XCDs 4-7 (options cluster_xcd_count / cluster_xcd_base; test.sh:9-12 runs its 23 jobs concurrently) — same
    models bit for bit, 1.4-1.8 x for such a pair (profiles/r05_config4_paired_anneals.txt); everything else runs as before."""
    import threading
    cids = list(mats)
    recs = []

    def collect(s, k, d10):
        cid, IF = cids[k], mats[cids[k]]
        if on_job:
            on_job(cid, s)
        x, e = s.coords(), s.energies()
        rho = s.score(IF)[2]                # K6 on the device, from the resident coordinates (= pipeline.spearman_IF_models(IF, x) to rounding)
        r = np.zeros((models, 5))
        r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4] = k, np.arange(models), e[:, 0], rho, s.last_timing()[0]
        recs.append(r)
        if out:
            d = os.path.join(out, cid)
            os.makedirs(d, exist_ok=True)
            pipeline.write_front_half(d10, d, f"{cid}_matrix")
            pipeline.assess_dgsa(d, f"{cid}_matrix", x, e, pipeline.restraints_from_dist10(d10), quiet=True)

    queue = list(mine)
    small = lambda k: mats[cids[k]].shape[0] <= PAIR_MAX_BEADS
    while queue:
        k = queue.pop(0)
        if second is not None and queue and small(k) and small(queue[0]):
            k2 = queue.pop(0)
            da = _prepare(solver, mats[cids[k]], models, seed, min_steps, gtol, 4)
            db = _prepare(second, mats[cids[k2]], models, seed, min_steps, gtol, 4)
            if solver.stat("cluster_ok") and second.stat("cluster_ok"):
                second.set_option("cluster_xcd_base", 4)
                err = []

                def other():
                    try:
                        second.run()
                    except Exception as ex:      # surfaces in the caller's thread
                        err.append(ex)
                th = threading.Thread(target=other)
                th.start()
                solver.run()
                th.join()
                if err:
                    raise err[0]
                collect(solver, k, da)
                collect(second, k2, db)
                continue
            queue.insert(0, k2)                   # no four-XCD geometry for one of them: one after the other, whole device
        d10 = _prepare(solver, mats[cids[k]], models, seed, min_steps, gtol, 8)
        solver.run()
        collect(solver, k, d10)
    order = {k: i for i, k in enumerate(mine)}
    recs.sort(key=lambda r: order[int(r[0, 0])])
    return np.concatenate(recs) if recs else np.zeros((0, 5))


def gather(rec, dist, device):
    """all_gather of record blocks of different length; identity without a process group."""
    if dist is None:
        return rec
    import torch
    world = dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rec.shape[0]], dtype=torch.int64, device=device))
    mmax = max(1, int(max(c.item() for c in counts)))
    buf = torch.zeros((mmax, rec.shape[1]), dtype=torch.float64, device=device)
    buf[: rec.shape[0]] = torch.from_numpy(rec).to(device)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    return np.concatenate([o[: int(c.item())].cpu().numpy() for o, c in zip(outs, counts)])


def rank_per_chromosome(rec, n_chrom):
    """Per chromosome: replica order by ascending int(E_noe), ties by replica id (chromosome3D.pl:796-802)."""
    out = []
    for k in range(n_chrom):
        r = rec[rec[:, 0] == k]
        out.append(r[np.lexsort((r[:, 1], r[:, 2].astype(np.int64)))])
    return out


def bench_block(s, rank, world, dist, device, sync_all=None, inputs=None, pair=True):
    """The `config4` block of a bench.py line: all 23 chromosomes at 500 kb x 20 replicas (test.sh:9-12; chr2_500kb is the documented
    stand-in), matrices to ranks by longest-processing-time-first on their restraint counts, every rank solves its share through the C
    ABI (full default schedule with the gradient exit, scoring included), ONE gather of the model records, per-chromosome ranking on rank
    0 — timed barrier to barrier around solve + gather (the matrices are parsed before: resident numpy arrays).  Returns the dict on rank
    0, None elsewhere."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sync_all = sync_all or (lambda: None)
    standins = set()
    mats = load_matrices(inputs or os.path.join(root, "tests", "golden", "all45"), "_500kb", standins)
    costs = job_costs(mats)
    mine = sharding.lpt_assign(costs, world)[rank]
    s2 = type(s)(s.device) if pair else None                # a second context on this rank's GPU: small chromosomes anneal in pairs
    solve_assigned(s, mats, mine[:1], 20)                  # first touch of this path (buffers of the largest job), untimed
    if s2 is not None:
        solve_assigned(s2, mats, mine[-1:], 20)
    sync_all()
    t0 = time.perf_counter()
    rec = solve_assigned(s, mats, mine, 20, second=s2)
    t_solve = time.perf_counter() - t0
    if s2 is not None:
        s2.close()
    allrec = gather(rec, dist, device)
    sync_all()
    wall = time.perf_counter() - t0
    load = np.array([[float(len(mine)), float(sum(costs[k] for k in mine)), t_solve, float(rec[:, 4].sum()) / 20.0]])
    loads = gather(load, dist, device)
    if rank != 0:
        return None
    per = rank_per_chromosome(allrec, len(mats))
    return {"workload": f"{len(mats)} chromosomes at 500 kb x 20 replicas ({len(allrec)} models), LPT over {world} rank(s), one gather",
            "wall_s": round(wall, 4), "models_per_s": round(len(allrec) / wall, 1), "standins": sorted(standins),
            "per_rank": [{"chromosomes": int(l[0]), "restraints": int(l[1]), "solve_s": round(float(l[2]), 4), "anneal_device_ms": round(float(l[3]), 2)} for l in loads],
            "paired_small_anneals": bool(pair), "chromosomes_ranked": len(per), "spearman_best_ranked_mean": round(-float(np.mean([r[0, 3] for r in per])), 4),
            "note": "paired_small_anneals: two consecutive chromosomes of <= 270 beads anneal side by side on disjoint halves of the GPU's XCDs (two contexts); "
                    "solve_s = the rank's wall for its chromosomes (K1, 5172-step schedule with gradient exit, read-back, Spearman of 20 models each); "
                    "wall_s = barrier to barrier incl. the gather; the reference runs this as 23 background processes (test.sh:9-12)"}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--inputs", default=os.path.join(root, "tests", "golden", "all45"))
    ap.add_argument("--pattern", default="_500kb")
    ap.add_argument("--models", type=int, default=20)
    ap.add_argument("--seed", type=int, default=82364)
    ap.add_argument("--min-steps", type=int, default=3000)
    ap.add_argument("--out", default=None, help="write the reference's output files per chromosome under this directory")
    ap.add_argument("--json", action="store_true", help="one JSON line with the per-chromosome ranking instead of the table")
    ap.add_argument("--gpus", type=int, default=0, help="ranks = GPUs (0: WORLD_SIZE if set, else 1); > 1 without WORLD_SIZE starts them")
    ap.add_argument("--dist", action="store_true", help="initialise the process group even at one rank (RCCL path on a one-GPU box)")
    ap.add_argument("--pair", type=int, default=1, help="1 (default): small chromosomes anneal in pairs on disjoint XCD halves of the rank's GPU; 0: one at a time")
    ap.add_argument("--bench-block", action="store_true", help="print bench.py's `config4` block (one JSON line) and nothing else")
    args = ap.parse_args(argv)

    from . import launch
    gpus = args.gpus if args.gpus > 0 else int(os.environ.get("WORLD_SIZE", "1"))
    rank, local, world = launch.ensure_ranks(gpus, sys.argv[1:] if argv is None else list(argv), module="chromosome3d_amd.batch",
                                             what="-m chromosome3d_amd.batch")
    dist, device, local = launch.init_process_group(local, world, force=args.dist)
    if args.bench_block:
        s = Solver(local)
        blk = bench_block(s, rank, world, dist, device, (lambda: dist.barrier()) if dist is not None else None, args.inputs, pair=bool(args.pair))
        s.close()
        if rank == 0:
            print(json.dumps(blk), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    t0 = time.perf_counter()
    standins = set()
    mats = load_matrices(args.inputs, args.pattern, standins)
    if not mats:
        sys.exit(f"no matrices matching {args.pattern!r} under {args.inputs}")
    mine = sharding.lpt_assign(job_costs(mats), world)[rank]
    s = Solver(local)
    s2 = Solver(local) if args.pair else None
    t1 = time.perf_counter()
    rec = solve_assigned(s, mats, mine, args.models, args.seed, args.min_steps, out=args.out, second=s2)
    t_solve = time.perf_counter() - t1
    s.close()
    if s2 is not None:
        s2.close()
    rec = gather(rec, dist, device)
    if rank == 0:
        per = rank_per_chromosome(rec, len(mats))
        if args.json:
            print(json.dumps({"world": world, "standins": sorted(standins), "chromosomes": {cid: {"order": [int(v) for v in r[:, 1]], "e_noe_int": [int(v) for v in r[:, 2]],
                                                                     "spearman_best": round(-float(r[0, 3]), 6)}
                                                               for cid, r in zip(mats, per)}}), flush=True)
        else:
            print(f"{len(mats)} chromosomes x {args.models} replicas on {world} rank(s); rank 0 solved {len(mine)} of them in {t_solve:.2f} s")
            for cid, r in zip(mats, per):
                print(f"  {cid:12s} N={mats[cid].shape[0]:4d} models={len(r):2d} best: replica {int(r[0, 1]):2d} E_noe={r[0, 2]:12.1f} "
                      f"Spearman(IF,1/d)={-r[0, 3]:.4f}  anneal {r[0, 4]:.1f} ms" + ("  [stand-in matrix]" if cid in standins else ""))
            print(f"  total wall incl. load/score {time.perf_counter() - t0:.2f} s; anneal device time summed over chromosomes "
                  f"{sum(float(r[0, 4]) for r in per):.1f} ms")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
