"""One process per GPU: how `bench.py --gpus N` and `python -m chromosome3d_amd.batch --gpus N` get their ranks.

The reference's batch is one OS process per matrix (`test.sh:4-12`); here it is one process per GPU under
`torch.distributed` (backend "nccl" = RCCL over xGMI).  Two ways in:

* the caller already started the ranks (`python -m torch.distributed.run --nproc-per-node N ... --gpus N`): WORLD_SIZE is
  set and must equal --gpus, else the run stops with a message instead of quietly measuring something else;
* plain `python bench.py --gpus N` with N > 1: this process starts the N ranks itself as CHILD processes
  (`torch.distributed.run`), before it has made any HIP or torch.cuda call, relays their output and exits with their status.
  It never replaces itself (no exec) and never touches the GPU: the GPUs are counted from the KFD topology in sysfs, torch is not
  even imported in the parent.

`--dist` (or C3D_BENCH_FORCE_DIST=1) makes a single process initialise the process group at world size 1, so that the
RCCL code path (init, device-side all_gather / all_reduce) also runs on a one-GPU box.
Rehearsal with fewer GPUs than ranks: C3D_BENCH_BACKEND=gloo (ranks share the devices there are).
"""
import os
import re
import socket
import subprocess
import sys

# Lines of the children's stdout that are library chatter, not output of ours: gloo's connection banner ("[Gloo] Rank 0 is connected to 1
# peer ranks ..."), c10d / torchrun log records ("[W1005 ...", "W1005 03:52:...", "[rank0]:[W...") and RCCL's "NCCL INFO/WARN" lines.
# Only these go to stderr; everything else a rank prints — rank 0's JSON line, the batch driver's per-chromosome table — is relayed on
# stdout, where `python -m chromosome3d_amd.batch --gpus N > out.txt` expects it (as the one-rank path prints it).
_CHATTER = re.compile(r"^\s*(\[Gloo\]|\[(W|E|I)\d{4} |(W|E|I)\d{4} \d\d:|\[rank\d+\]:\[|[\w.-]+:\d+:\d+ \[\d+\] NCCL |NCCL (INFO|WARN))")


def is_chatter(line):
    return bool(_CHATTER.match(line))


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _index_list(value):
    return [t for t in value.replace(" ", "").split(",") if t != ""]


def visible_gpus(root="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs a child process will be able to use, counted WITHOUT loading HIP, HSA or torch: the KFD topology nodes with SIMDs
    (`simd_count > 0` in <node>/properties; CPU nodes have 0), capped by what ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES leave visible.  Returns None when the topology cannot be read (no amdgpu driver in this container, or a
    sandbox without sysfs): the caller then lets the children find out."""
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as fh:
                for line in fh:
                    parts = line.split()
                    if len(parts) == 2 and parts[0] == "simd_count" and int(parts[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            n = min(n, len(_index_list(os.environ[var])))
    return n


def ensure_ranks(gpus, argv, script=None, module=None, what="bench.py"):
    """Returns (rank, local_rank, world) for this process, starting the ranks first when that is this process's job
    (in which case it does not return: it exits with the children's status)."""
    if gpus < 1:
        sys.exit(f"{what}: --gpus must be >= 1")
    ws = os.environ.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            sys.exit(f"{what}: --gpus {gpus} but WORLD_SIZE={ws}: start it as `python -m torch.distributed.run --nproc-per-node {gpus} ... "
                     f"--gpus {gpus}`, or as plain `python {what} --gpus {gpus}` without WORLD_SIZE in the environment")
        return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(ws)
    if gpus == 1:
        return 0, 0, 1
    backend = os.environ.get("C3D_BENCH_BACKEND", "nccl")
    have = visible_gpus()
    if backend == "nccl" and have is not None and have < gpus:
        sys.exit(f"{what}: --gpus {gpus} asked for, this machine exposes {have} GPU(s): one rank per GPU over RCCL needs {gpus}. "
                 f"(C3D_BENCH_BACKEND=gloo rehearses the {gpus}-rank code path on the GPUs there are; its timings mean nothing.)")
    target = ["-m", module] if module else [script]
    # --standalone: torch.distributed.run picks a free rendezvous port itself (no bind / close / reuse race between concurrent launches)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(gpus)] + target + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                                # what the ranks print comes through here; library chatter goes to stderr
        out = sys.stderr if is_chatter(line) else sys.stdout
        out.write(line)
        out.flush()
    sys.exit(proc.wait())


def init_process_group(local_rank, world, force=False):
    """(dist module or None, collective device or None, local device index).  Imports torch FIRST when a group is needed:
    its bundled HIP runtime becomes the one libc3d.so binds to."""
    force = force or os.environ.get("C3D_BENCH_FORCE_DIST", "0") not in ("", "0")
    if world == 1 and not force:
        return None, None, local_rank
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:                                         # forced group of one: nobody started us through torch.distributed.run
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    backend = os.environ.get("C3D_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < max(world, 1):
        # (the parent counts GPUs from sysfs; where that is unreadable the ranks find out here)
        sys.exit(f"--gpus {world} asked for, this machine exposes {ndev} GPU(s): one rank per GPU over RCCL needs {world}. "
                 f"(C3D_BENCH_BACKEND=gloo rehearses the {world}-rank code path on the GPUs there are; its timings mean nothing.)")
    local = local_rank % max(ndev, 1)
    if ndev:
        torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return dist, ("cuda" if backend == "nccl" else "cpu"), local
