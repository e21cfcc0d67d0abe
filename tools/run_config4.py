"""BASELINE.json configs[3]: every 500 kb chromosome x 20 replicas, sharded over the GPUs of a node.

    python tools/run_config4.py                                   # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_config4.py
Chromosomes are assigned to ranks by longest-processing-time-first on their restraint counts
(sharding.lpt_assign); every rank solves its chromosomes with 20 replicas each; one all_gather of the
per-model records (RCCL on GPUs; C3D_BENCH_BACKEND=gloo for a rehearsal with fewer GPUs than ranks)
brings everything to rank 0, which ranks the models per chromosome as chromosome3D.pl:796-802 does.
Needs tests/golden/all45 (tools/pack_all_inputs.py).  chr2_500kb is missing upstream (.MISSING_LARGE_BLOBS).
"""
import glob, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
dist = None
if world > 1:
    import torch, torch.distributed as dist
    backend = os.environ.get("C3D_BENCH_BACKEND", "nccl")
    local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline, sharding

ALL = os.path.join(ROOT, "tests", "golden", "all45")
cids = sorted((os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_500kb_upper.npz")),
              key=lambda c: int(re.match(r"chr(\d+)", c).group(1)))
def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m
mats = {c: load(c) for c in cids}
cost = [(mats[c].shape[0] - 5) * (mats[c].shape[0] - 4) // 2 for c in cids]      # ~ restraints
mine = sharding.lpt_assign(cost, world)[rank]
t0 = time.perf_counter()
s = Solver(local)
recs = []
for k in mine:
    cid, IF = cids[k], mats[cids[k]]
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), None, 1e-2, 250)
    s.init_replicas(20, 82364, 0); s.run()
    x, e = s.coords(), s.energies()
    rho = pipeline.spearman_IF_models(IF, x)
    r = np.zeros((20, 5)); r[:, 0] = k; r[:, 1] = np.arange(20); r[:, 2] = e[:, 0]; r[:, 3] = rho; r[:, 4] = s.last_timing()[0]
    recs.append(r)
rec = np.concatenate(recs) if recs else np.zeros((0, 5))
t_solve = time.perf_counter() - t0
if dist is not None:
    import torch
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev))
    mmax = int(max(c.item() for c in counts))
    buf = torch.zeros((mmax, 5), dtype=torch.float64, device=dev); buf[: rec.shape[0]] = torch.from_numpy(rec).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    rec = np.concatenate([o[: int(c.item())].cpu().numpy() for o, c in zip(outs, counts)])
if rank == 0:
    print(f"config 4: {len(cids)} chromosomes x 20 replicas on {world} rank(s); rank 0 solved {len(mine)} in {t_solve:.2f} s")
    for k, cid in enumerate(cids):
        r = rec[rec[:, 0] == k]
        best = r[np.lexsort((r[:, 1], r[:, 2].astype(np.int64)))][0]
        print(f"  {cid:12s} N={mats[cid].shape[0]:4d} models={len(r):2d} best: replica {int(best[1]):2d} E_noe={best[2]:12.1f} Spearman(IF,1/d)={-best[3]:.4f}  anneal {best[4]:.1f} ms")
    print(f"  total wall incl. load/score {time.perf_counter() - t0:.2f} s")
if dist is not None:
    dist.barrier(); dist.destroy_process_group()
