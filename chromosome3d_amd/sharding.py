"""Replica sharding across GPUs and the final gather for ranking.

Replicas are independent: there is no per-step exchange.  The only collective is one
all_gather of fixed-size records after the solve, so that rank 0 can rank all models exactly
as chromosome3D.pl:796-802 does.  torch.distributed is plumbing here ("nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests); records are latency-bound (12 N + 16 bytes each).
"""
import numpy as np


def replica_range(total, world, rank):
    """Contiguous block of replica ids for `rank` when `total` replicas are split over `world`
    ranks as evenly as possible (20 over 8 -> 3,3,3,3,2,2,2,2)."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def lpt_assign(costs, world):
    """Longest-processing-time-first assignment of jobs (cost ~ restraints x steps) to ranks;
    returns a list of job-index lists.  Used for the all-chromosomes batch (config 4)."""
    order = sorted(range(len(costs)), key=lambda k: (-costs[k], k))
    load = [0.0] * world
    out = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        out[r].append(k)
        load[r] += costs[k]
    return out


def pack_records(replica_ids, e_noe, spearman, xyz):
    """[M, 4 + 3N] float64 records: id, E_noe, spearman, n, xyz..."""
    M, n = xyz.shape[0], xyz.shape[1]
    rec = np.zeros((M, 4 + 3 * n), dtype=np.float64)
    rec[:, 0] = replica_ids
    rec[:, 1] = e_noe
    rec[:, 2] = spearman
    rec[:, 3] = n
    rec[:, 4:] = xyz.reshape(M, -1)
    return rec


def gather_records(rec, device=None):
    """all_gather of per-rank record blocks (possibly different M per rank).  Returns the
    concatenation ordered by replica id on every rank.  Without an initialised process group
    this is the identity."""
    import sys
    if "torch" not in sys.modules:          # single process: nothing to gather, do not pull torch in
        return rec[np.argsort(rec[:, 0], kind="stable")]
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rec[np.argsort(rec[:, 0], kind="stable")]
    # an initialised group of ONE rank still goes through the collectives (bench.py --dist: the RCCL path on a one-GPU box)
    world = dist.get_world_size()
    dev = device if device is not None else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev))
    mmax = int(max(c.item() for c in counts))
    buf = torch.zeros((mmax, rec.shape[1]), dtype=torch.float64, device=dev)
    buf[: rec.shape[0]] = torch.from_numpy(rec).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    parts = [o[: int(c.item())].cpu().numpy() for o, c in zip(outs, counts)]
    allrec = np.concatenate(parts, axis=0)
    return allrec[np.argsort(allrec[:, 0], kind="stable")]


def rank_models(rec):
    """Indices into rec ordered by ascending int(E_noe), ties by replica id (:796-802)."""
    return sorted(range(rec.shape[0]), key=lambda k: (int(rec[k, 1]), rec[k, 0]))
