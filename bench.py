#!/usr/bin/env python3
"""Headline benchmark: SA steps/s for 20 replicas of chr1_500kb (N = 455 beads, R = 101426
restraints) per MI355X — BASELINE.json configs[2], the configuration the metric is quoted on.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reps R] [--scaling weak|strong] [--dist]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts the N ranks itself
(child processes under torch.distributed.run, before any HIP call) and relays rank 0's line; with WORLD_SIZE set it must
equal N.  `--dist` initialises the RCCL group even at one rank.  (chromosome3d_amd/launch.py)

A "step" is one SA step (force evaluation + coordinate update) of every replica on the GPU.
The timed region is exactly K steps of the real annealing schedule, starting W steps in, bracketed
by a barrier + device synchronisation on both sides (a rank's clock stops when c3d_run_steps returns: the multi-step launch has written
its completion mark — the host watches it for `spin_wait_us`, then falls back to hipStreamSynchronize —; the device-wide synchronise and
the closing barrier follow, and the region's time is the maximum over the ranks; `timing.value_barrier_to_barrier` is the jointly
bracketed clock of the same regions).  The bracket is repeated `reps` times
(consecutive K-step regions of the same schedule, the next batch of replicas starting when the
schedule ends); `value` comes from the MEDIAN region wall time, maximum over the ranks.

Nothing is built inside a timed region: one untimed pass replays the identical call pattern first
(hipGraphs of the per-step path are captured there; the multi-step cluster kernel needs none), and
the library's `graph_captures` counter is asserted unchanged over the timed regions.

--scaling strong (default for --gpus N > 1) 20 replicas IN ALL, split 3,3,3,3,2,2,2,2 over the ranks (sharding.replica_range):
                 the north star's "20 replicas of chr1_500kb at 1/2/4/8 MI355X"; the weak figure of the same run rides along
--scaling weak   (the one-GPU default, where the two coincide) every rank runs its own 20 replicas (ids rank*20 ..)
Every line with more than one rank — and the one-GPU line — also carries `config4`: all 23 chromosomes at 500 kb x 20 replicas,
matrices to ranks by LPT, one gather (python -m chromosome3d_amd.batch's code), wall-clock barrier to barrier with the per-rank load;
the one-GPU line carries `end_to_end`: the user-facing path of chromosome3D.pl for chr1_500kb as a child process (c3d_batch).
After the timed regions one all_gather (RCCL) of the per-replica records of a COMPLETE untimed anneal
lets rank 0 rank all models (reported: gather_ms, spearman_*).

Output: ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, algorithmic
bytes B = 4R + 72N per replica-step; duration from HIP events on the solver's stream) and
`cpu_baseline` (the fp64 oracle on the host cores over a bounded sample of the same workload).
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REPLICAS = 20
WORKLOAD = "chr1_500kb"
MIN_STEPS = 3000
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def cpu_baseline(IF, d10, model, fire, stages, budget_s=8.0):
    """Oracle (fp64 C) on a bounded sample: one replica on one core, as many SA steps of the same
    schedule as fit in ~budget_s; then the same sample on every host core at once."""
    from oracle import oracle as O
    from tests.util import oracle_fire_from, oracle_model_from
    n = IF.shape[0]
    om, of = oracle_model_from(model, n), oracle_fire_from(fire)
    rows = [(s.kind, s.nsteps, s.dt, s.w_all, s.w_vdw, s.repel_s, s.t_bath) for s in stages]
    # calibrate on the pre-minimisation stage, then time a slice of hot + cool stages
    t0 = time.perf_counter()
    x, v, ev0 = O.run_schedule(om, d10, O.make_stages(rows[:1]), of, 82364, 0)
    per = (time.perf_counter() - t0) / ev0
    budget_steps = max(200, int(budget_s / per))
    sample, total = [], 0
    for r in rows[1:]:
        if total >= budget_steps:
            break
        take = min(r[1], budget_steps - total)
        sample.append((r[0], take) + tuple(r[2:]))
        total += take
    t0 = time.perf_counter()
    _, _, ev = O.run_schedule(om, d10, O.make_stages(sample), of, 82364, 0, x0=x)
    dt = time.perf_counter() - t0
    kinds = "+".join(sorted({{0: "hot MD", 1: "cool MD", 2: "FIRE", 5: "final minimisation (two-point steps)"}[r[0]] for r in sample}))
    one = (ev / dt, f"1 replica of {WORKLOAD}, {ev} SA steps of the same schedule after its first stage ({kinds}), fp64 oracle/c3d_oracle.c, "
                    f"1 core, {dt:.1f} s")
    # the same sample on every host core at once, one replica per thread (the reference's way to use a CPU box is one
    # process per chromosome, test.sh:4-12); ctypes releases the GIL inside the C call
    import threading
    T = max(1, min(16, os.cpu_count() or 1))
    evs = [0] * T

    def work(k):
        evs[k] = O.run_schedule(om, d10, O.make_stages(sample), of, 82364, 100 + k, x0=x)[2]

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    all_cores = {"value": round(sum(evs) / dt, 1), "unit": "replica-steps/s", "cores": T,
                 "sample": f"{T} replicas at once, one thread each, {sum(evs)} SA steps, {dt:.1f} s"}
    return one[0], one[1], all_cores


def measured_stream_peaks():
    """The HBM stream rates measured on this machine type (tools/microbench/hbm_stream.hip), from the newest
    profiles/*hbm_stream*.txt: {"read": GB/s, "copy": GB/s, "triad": GB/s, "source": file} or None."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_stream*.txt")))
    if not files:
        return None
    out = {}
    for line in open(files[-1]):
        mm = re.match(r"(read|copy|triad)\s.*best (\d+) GB/s", line)
        if mm:
            out[mm.group(1)] = float(mm.group(2))
    if not out:
        return None
    out["unit"] = "GB/s"
    out["source"] = os.path.relpath(files[-1], ROOT)
    return out


def hbm_traffic_from_profiles(kernel, n, replicas):
    """HBM bytes per SA step of `kernel` from the newest profiles/*hbm_traffic*.json that matches (written by
    tools/pmc_summarise.py from separate rocprofv3 --pmc passes); None when there is none."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic*.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("kernel", "").split("<")[0] == kernel.split("<")[0] and rec.get("n") == n and rec.get("replicas") == replicas:
            best = (rec.get("hbm_bytes_per_sa_step"), os.path.relpath(path, ROOT))
    return best if best else (None, None)


def side_figures(device, IF, model, fire, stages, args, B):
    """Driver-visible second figures (one GPU, rank 0): the precision-matched fp64 leg on the headline workload (the
    reference's CNS arithmetic is fp64; c3d_f64.hip is the oracle's algorithm in its precision on the GPU) and the
    step rates of BASELINE configs[1] (chr21_1mb x 20) and configs[4] (synthetic N = 2500 x 8), each from the HIP-event
    pair around c3d_run_steps on the solver's stream over a whole untimed-by-the-metric anneal."""
    from chromosome3d_amd import Solver, default_schedule, pipeline
    from tests.util import load_if, synthetic_if
    out = {}
    s = Solver(device)
    try:
        # ---- fp64 leg: same workload over the SAME stretch of the schedule as the fp32 line's regions at the driver's arguments (W warm-up
        #      steps, then 1000 steps: 50 regions of 20 there, 5 regions of 200 here — a 20-step region of the per-step path is a third
        #      launch-and-join overhead); 3 untimed passes (graphs, clocks), then the median of 5 ----
        s.set_option("precision", 64)
        s.set_model(model)
        pipeline.IF2dist_new(s, IF)
        s.set_schedule(stages, fire, 0.0, 250)
        K, NREG = 200, 5
        passes = []
        for rep in range(8):
            s.init_replicas(REPLICAS, 82364, 0)
            s.run_steps(max(args.warmup, 1))
            wall = dev_ms = 0.0
            did = 0
            per = []
            for _ in range(NREG):
                t0 = time.perf_counter()
                d = s.run_steps(K)
                w = time.perf_counter() - t0
                wall += w; dev_ms += s.last_timing()[0]; did += d
                per.append(round(1e6 * w / max(d, 1), 3))
            if rep >= 3:
                passes.append((wall, dev_ms, did, per))
        wall, dev_ms, did, per = sorted(passes)[len(passes) // 2]
        v64 = REPLICAS * did / wall
        out["value_f64"] = round(v64, 1)
        out["f64"] = {"value": round(v64, 1), "unit": "replica-steps/s", "steps": did, "regions": NREG, "us_per_step_wall_by_region": per,
                      "us_per_step_device": round(1e3 * dev_ms / did, 3),
                      "frac_f64": round(REPLICAS * B / (1e-3 * dev_ms / did) / 1e9 / HBM_PEAK_GBS, 4), "kernel": s.step_kernel_name,
                      "note": "option precision=64: one k64_step launch per SA step of a replica group (hipGraph replay, two groups on two streams), "
                              "the oracle's algorithm in fp64 (the reference's precision); same B per replica-step, device time from the HIP-event "
                              "pair on the solver's stream; the regions cover steps W .. W + 1000 of the schedule (195 minimiser steps, then hot MD), "
                              "the stretch the fp32 line's 50 regions of 20 steps cover at the driver's arguments"}
        out["frac_f64"] = out["f64"]["frac_f64"]
    finally:
        s.close()
    s = Solver(device)
    try:
        other = {}
        for key, name, mat, nrep, nmin in (("config2", "chr21_1mb x 20", load_if("chr21_1mb"), 20, MIN_STEPS),
                                           ("config5", "synthetic N=2500 x 8 (SURVEY 8d recipe)", synthetic_if(2500)[0], 8, 1000)):
            n = mat.shape[0]
            s.set_model(model)
            pipeline.IF2dist_new(s, mat)
            s.set_schedule(default_schedule(nmin), fire, 0.0, 250)
            for _ in range(2):                                         # first pass builds whatever the path needs
                s.init_replicas(nrep, 82364, 0)
                s.run_steps(10 ** 7)
            ms, steps, la = s.last_timing()
            Bc = 4 * s.num_restraints + 72 * n
            other[key] = {"workload": name, "n": n, "restraints": s.num_restraints, "replicas": nrep, "sa_steps": steps, "launches": la,
                          "us_per_step_device": round(1e3 * ms / steps, 3), "replica_steps_per_s": round(nrep * steps / (1e-3 * ms), 1),
                          "roofline_frac": round(nrep * Bc / (1e-3 * ms / steps) / 1e9 / HBM_PEAK_GBS, 4), "kernel": s.step_kernel_name}
        out["other_configs"] = other
    finally:
        s.close()
    return out


def config4_in_a_child():
    """One GPU: BASELINE configs[3] through `python -m chromosome3d_amd.batch --bench-block` as a child process of its own (not under
    whatever profiler watches this one: its chr1_500kb job runs the headline kernel instantiation in launches of another length, and the
    kernel trace of THIS command must show the timed launches only).  More ranks: in process, batch.bench_block (below)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS")) and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--bench-block"], capture_output=True, text=True, cwd=ROOT, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stdout + p.stderr)[-300:]}
    return json.loads(lines[-1])


def final_stage_block(device=0):
    """The final minimisation's two kernels at the headline workload, each as ONE launch with its own start-to-end stamps: the 1000
    two-point steps on k_cluster_tp and the FIRE steps behind them on k_cluster (median of 3 anneals).  The timed regions of the metric
    stay in front of that stage (config.L_timed), so its kernels get figures of their own — measured in a child process
    (`bench.py --final-stage-block`), not under whatever profiler watches the parent: the parent's kernel trace must hold the timed
    launches only, or its average duration of k_cluster would not be roofline.avg_launch_us."""
    from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
    from tests.util import load_if
    s = Solver(device)
    try:
        s.set_model(default_model())
        pipeline.IF2dist_new(s, load_if(WORKLOAD))
        s.set_schedule(default_schedule(MIN_STEPS), default_fire(), 0.0, 250)
        L = s.schedule_length
        s.set_option("kernel_timing", 1)
        tp, fr, md, names = [], [], [], {}
        for rep in range(4):
            s.init_replicas(REPLICAS, 82364, 10 ** 6)
            s.run_steps(L - MIN_STEPS)
            md.append(s.stat("last_kernel_us") / (L - MIN_STEPS))
            s.run_steps(1000)
            tp.append(s.stat("last_kernel_us") / 1000.0)
            names["two_point"] = s.step_kernel_name
            s.run_steps(MIN_STEPS - 1000)
            fr.append(s.stat("last_kernel_us") / (MIN_STEPS - 1000))
            names["fire"] = s.step_kernel_name
        med = lambda v: round(statistics.median(v[1:]), 4)            # the first anneal warms the clocks
        return {"k_cluster_tp": {"kernel": names["two_point"], "us_per_step_kernel": med(tp), "steps_per_launch": 1000,
                                 "range": f"steps {L - MIN_STEPS} .. {L - MIN_STEPS + 1000} of the schedule (two-point step sizes)"},
                "k_cluster_fire_part": {"kernel": names["fire"], "us_per_step_kernel": med(fr), "steps_per_launch": MIN_STEPS - 1000,
                                        "range": f"steps {L - MIN_STEPS + 1000} .. {L} (FIRE after the hand-over)"},
                "k_cluster_before_final_stage": {"us_per_step_kernel": med(md), "steps_per_launch": L - MIN_STEPS,
                                                 "range": f"steps 0 .. {L - MIN_STEPS} in one launch (FIRE 200, hot MD 1000, cool MD 972)"},
                "note": f"{WORKLOAD} x {REPLICAS}, one GPU, kernel start-to-end stamps of one launch each, median of 3 anneals; a child process; not part of `value`"}
    finally:
        s.close()


def final_stage_in_a_child():
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS")) and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--final-stage-block"], capture_output=True, text=True, cwd=ROOT, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stdout + p.stderr)[-300:]}
    return json.loads(lines[-1])


def end_to_end(IF):
    """The user-facing path of chromosome3D.pl for the headline matrix, as a child process: c3d_batch on chr1_500kb's text matrix —
    parse -> K1 -> <ID>.dist/.rr/contact.tbl/.fasta -> 20 start structures -> anneal -> read-back, ranking, Spearman -> 20 PDB files,
    the reference's satisfaction table (c3d_assess) -> shaping (filter_nonCA/reindex/CONECT) -> <ID>_model1..5.pdb.  The text matrix is
    written here first (untimed: the reference gets it as its input)."""
    import re
    import subprocess
    import tempfile
    from tests.util import write_if_text
    exe = os.path.join(ROOT, "chromosome3d_amd", "_lib", "c3d_batch")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as td:
        mat = os.path.join(td, f"{WORKLOAD}_matrix.txt")
        write_if_text(IF, mat)
        runs = []
        for _ in range(3):
            t0 = time.perf_counter()
            # (a child of its own: not under whatever profiler watches this process)
            env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
            p = subprocess.run([exe, mat, "--out", os.path.join(td, "out"), "--lanes", "1", "-m", str(REPLICAS)], capture_output=True, text=True, env=env)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": (p.stdout + p.stderr)[-300:]}
            mm = re.search(r"end-to-end ([0-9.]+) s.*\[phases: parse\+K1 ([0-9.]+), front-half files\+start structures ([0-9.]+), anneal ([0-9.]+), "
                           r"read-back\+rank\+Spearman ([0-9.]+), PDB\+assessment\+shaping ([0-9.]+) s\]", p.stdout)
            runs.append((wall, [float(g) for g in mm.groups()] if mm else None))
        wall, ph = sorted(runs, key=lambda r: r[0])[1]
        out = {"workload": f"{WORKLOAD} x {REPLICAS} models, c3d_batch --lanes 1 as a child process (median of 3)", "process_wall_s": round(wall, 3)}
        if ph:
            out.update({"job_s": ph[0], "phases_s": {"parse_and_K1": ph[1], "front_half_files_and_start_structures": ph[2], "anneal": ph[3],
                                                      "read_back_rank_spearman": ph[4], "pdb_assessment_shaping": ph[5]},
                        "process_start_and_device_init_s": round(wall - ph[0], 3)})
        # the reference's own command line (bin/chromosome3D_amd.pl: Perl -> XS or job.sh -> libc3d), which also writes what the reference's
        # assess_dgsa writes in Perl — contact_violation.txt alone is 20 x 101 426 rows
        import shutil
        if shutil.which("perl"):
            walls = []
            for k in range(2):      # the first run on a fresh box also pages perl and its modules in from the image; both are reported
                t0 = time.perf_counter()
                p = subprocess.run(["perl", os.path.join(ROOT, "bin", "chromosome3D_amd.pl"), "-i", mat, "-o", os.path.join(td, f"perl_out{k}"), "-m", str(REPLICAS)],
                                   capture_output=True, text=True, env=env)
                walls.append(round(time.perf_counter() - t0, 2) if p.returncode == 0 else None)
            out["perl_driver_wall_s"] = walls[1]
            out["perl_driver_first_run_s"] = walls[0]
            out["perl_driver_note"] = ("perl bin/chromosome3D_amd.pl -i <matrix> -o <dir> -m 20: the reference's CLI and every file it leaves; the satisfaction "
                                       "table and contact_violation.txt (2.03 M rows) come from the library (c3d_write_violations through the XS binding)")
        out["reference_recorded"] = ("chromosome3D.pl on this matrix, measured in the build container on one core (BASELINE.md 2), NOT on this box: Perl front half "
                                     "3.3 s + assessment 4.5 s per model (90 s for 20); its CNS leg cannot run here")
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=172)
    ap.add_argument("--reps", type=int, default=0, help="timed K-step regions (0 = 50 for K <= 100, else 5)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None, help="default: strong for --gpus N > 1 (20 replicas in all), weak at one GPU")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32",
                    help="f64: the fp64 reference step (c3d_f64.hip, written for clarity) instead of the fp32 product kernels")
    ap.add_argument("--dist", action="store_true", help="initialise the torch.distributed process group even at one rank (RCCL path on a one-GPU box)")
    ap.add_argument("--no-side-figures", action="store_true", help="skip the f64 leg and the config 2 / config 5 step rates (rank 0, one GPU only)")
    ap.add_argument("--final-stage-block", action="store_true", help="internal: print the final stage's kernel figures as one JSON line and exit (final_stage_in_a_child)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicas", type=int, default=REPLICAS, help="weak: per GPU; strong: in all")
    ap.add_argument("--groups", type=int, default=0, help="replica groups on separate streams, per-step path (0 = library default)")
    ap.add_argument("--resident", type=int, default=-1, help="1/0: multi-step cluster kernel on/off (-1 = library default)")
    ap.add_argument("--rpw", type=int, default=0, help="rows per wave of the step kernel (tuning knob; 0 = library default)")
    args = ap.parse_args()
    reps = args.reps or (50 if args.steps <= 100 else 5)
    if args.final_stage_block:
        print(json.dumps(final_stage_block()), flush=True)
        return

    # --gpus N: either the caller started N ranks (WORLD_SIZE == N, checked) or this process starts them as children and
    # relays their output — before anything here has touched HIP (chromosome3d_amd/launch.py)
    from chromosome3d_amd import launch
    rank, local_rank, world = launch.ensure_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__), what="bench.py")
    # torch (when a group is needed) is imported in there, before libc3d.so: its bundled HIP runtime is the one both bind to.
    # C3D_BENCH_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
    dist, coll_dev, local_rank = launch.init_process_group(local_rank, world, force=args.dist)
    torch = sys.modules.get("torch")
    on_gpu_group = coll_dev == "cuda"
    if args.scaling is None:                      # the north star's case is "20 replicas at 1/2/4/8 GPUs": strong scaling
        args.scaling = "strong" if world > 1 else "weak"

    from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline, sharding
    from tests.util import load_if

    IF = load_if(WORKLOAD)
    n = IF.shape[0]
    if args.scaling == "strong":
        first, M = sharding.replica_range(args.replicas, world, rank)      # 20 over 8 -> 3,3,3,3,2,2,2,2
        total_replicas = args.replicas
    else:
        M, first, total_replicas = args.replicas, rank * args.replicas, args.replicas * world
    s = Solver(local_rank)
    if args.dtype == "f64":
        s.set_option("precision", 64)
    model, fire, stages = default_model(), default_fire(), default_schedule(MIN_STEPS)
    s.set_model(model)
    d10 = pipeline.IF2dist_new(s, IF)            # K1 on the GPU; targets stay resident in HBM
    R = s.num_restraints
    s.set_schedule(stages, fire, 0.0, 250)       # gtol 0: fixed-length schedule (no early exit)
    s.set_option("use_graph", 0 if args.no_graph else 1)
    if args.rpw:
        s.set_option("rows_per_wave", args.rpw)
    if args.groups:
        s.set_option("replica_groups", args.groups)
    if args.resident >= 0:
        s.set_option("resident", args.resident)
    L = s.schedule_length
    L_timed = L - MIN_STEPS if args.steps <= L - MIN_STEPS else L       # where the timed regions end (see pattern)
    totals = {"sa_steps": 0, "launches": 0}

    def sync_all():
        if dist is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    # ---- one COMPLETE anneal, untimed by the metric: models for the parity keys, wall-clock per chromosome.  It runs in
    #      calls of K steps like the timed regions, so that every launch of the step kernel in this process covers (about)
    #      K SA steps and the kernel's average duration in a rocprofv3 --kernel-trace --stats of this command is the
    #      number `roofline.avg_launch_us` reports ----
    def full_anneal(first_id):
        s.init_replicas(M, 82364, first_id)
        dev = 0.0
        while s.steps_done < L:
            totals["sa_steps"] += s.run_steps(min(args.steps, L - s.steps_done))
            ms, _, la = s.last_timing()
            dev += ms
            totals["launches"] += la
        return dev
    full_anneal(10 ** 6 + first)                  # first pass: builds whatever the full schedule needs
    t0 = time.perf_counter()
    full_dev_ms = full_anneal(first)
    full_wall = time.perf_counter() - t0
    xyz = s.coords()
    en = s.energies()
    xyz = xyz - xyz.mean(axis=1, keepdims=True)

    # ---- the benchmark's call pattern: init, W warm-up steps, `reps` regions of K steps -----------------------
    def pattern(timed, M=M, first=first, total_replicas=total_replicas):
        # the schedule position is tracked here, so that inside the clock there is nothing but c3d_run_steps
        batch, pos = 0, 0
        s.init_replicas(M, 82364, first)
        left_w = args.warmup
        while left_w > 0:
            if pos >= L:
                batch += 1
                pos = 0
                s.init_replicas(M, 82364, batch * total_replicas + first)
            did = s.run_steps(min(left_w, L - pos))
            left_w -= did
            pos += did
            totals["sa_steps"] += did
            totals["launches"] += s.last_timing()[2]
        walls, devs, kerns, launches = [], [], [], 0
        bbs.clear()
        for _ in range(reps):
            # a region that would cross the end of the schedule starts a fresh batch of replicas first (untimed); since the end of round 5
            # also one that would enter the final minimisation, if a region fits before it: the metric is SA steps, and that stage's steps
            # run in kernels of their own (k_cluster_tp for its two-point part, then k_cluster again: two launches where a region held both)
            if pos + args.steps > L_timed and args.steps <= L_timed:
                batch += 1
                pos = 0
                s.init_replicas(M, 82364, batch * total_replicas + first)
            if timed:
                sync_all()
            t_open = time.perf_counter()          # this rank has left the opening bracket
            wall, left, dev, kern = 0.0, args.steps, 0.0, 0.0
            while left > 0:
                if pos >= L:                                              # only when --steps exceeds the schedule
                    batch += 1
                    pos = 0
                    s.init_replicas(M, 82364, batch * total_replicas + first)
                want = min(left, L - pos)
                t0 = time.perf_counter()
                did = s.run_steps(want)                                   # returns when the range is done (completion mark / stream synchronise)
                wall += time.perf_counter() - t0
                left -= did
                pos += did
                ms, _, la = s.last_timing()
                dev += ms
                kern += s.stat("last_kernel_us")
                launches += la
                totals["sa_steps"] += did
                totals["launches"] += la
            if timed and dist is not None:
                # closing bracket.  A rank's clock has stopped when c3d_run_steps returned: the last workgroup of the multi-step launch has
                # written its completion mark into host-mapped memory (c3d_api.cpp run_cluster; ~6 us before the stream retires the
                # kernel — one-rank regions follow each other back to back, so whatever drains after the mark delays the NEXT region's
                # launch and cannot hide) — the only stream of this process with work in the region, and exactly what the one-rank line
                # measures; the device-wide
                # torch.cuda.synchronize() (~4 us on an idle device) and the barrier (tens of microseconds over RCCL: a third of
                # a 20-step region) follow.  The region's time is the MAXIMUM over the ranks (all_reduce below): the slowest
                # rank counts, the bracket's own latency does not.
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
                dist.barrier()
            bbs.append(time.perf_counter() - t_open)    # barrier to barrier: the jointly bracketed region (one rank: = the rank's clock + loop bookkeeping)
            walls.append(wall)
            devs.append(dev)
            kerns.append(kern)
        return walls, devs, kerns, launches

    bbs = []
    pattern(False)                                # untimed: every graph the pattern needs exists afterwards
    cap0 = s.stat("graph_captures")
    fb0 = s.stat("resident_fallbacks")
    s.set_option("event_timing", 0)               # nothing but launch + synchronise inside the clock (an event pair costs 2-5 us per call)
    walls, _, _, launches = pattern(True)
    bb_walls = list(bbs)
    s.set_option("event_timing", 1)
    captures_in_timed = s.stat("graph_captures") - cap0
    fallbacks_in_timed = s.stat("resident_fallbacks") - fb0
    # the same regions once more, untimed by the metric, for the device-side durations: the HIP-event pair around each call
    # and the multi-step kernel's own start/end stamps (hipExtLaunchKernel events on the solver's stream; ~15 us of host
    # time per launch, so not above)
    s.set_option("kernel_timing", 1)
    _, devs, kerns, _ = pattern(False)
    s.set_option("kernel_timing", 0)
    assert captures_in_timed == 0, f"{captures_in_timed} hipGraph captures inside the timed regions"
    kernel = s.step_kernel_name
    path = int(s.stat("last_path"))

    if dist is not None:
        t = torch.tensor([walls, devs, kerns, bb_walls], dtype=torch.float64, device="cuda" if on_gpu_group else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                       # per region: the slowest rank
        walls, devs, kerns, bb_walls = t[0].tolist(), t[1].tolist(), t[2].tolist(), t[3].tolist()
        fw = torch.tensor([full_wall, full_dev_ms, float(M)], dtype=torch.float64, device="cuda" if on_gpu_group else "cpu")
        allfw = [torch.zeros_like(fw) for _ in range(world)]
        dist.all_gather(allfw, fw)
        full_wall = max(float(a[0]) for a in allfw)
        full_dev_ms = max(float(a[1]) for a in allfw)
        per_rank = [int(a[2]) for a in allfw]
    else:
        per_rank = [M]
    wall = statistics.median(walls)
    bb_wall = statistics.median(bb_walls)
    dev_ms = statistics.median(devs)
    kern_us = statistics.median(kerns)             # 0 on the per-step paths (no single kernel to stamp)

    # ---- scoring + the one collective (not timed) ----
    rho = pipeline.spearman_IF_models(IF, xyz) if M > 0 else np.zeros(0)
    rec = sharding.pack_records(first + np.arange(M), en[:, 0], rho, xyz)
    tg = time.perf_counter()
    allrec = sharding.gather_records(rec, device="cuda" if on_gpu_group else None)
    gather_ms = 1e3 * (time.perf_counter() - tg)
    order = sharding.rank_models(allrec)

    # strong mode: the weak-scaling figure of the same job size per GPU rides along as a second key
    weak_value = None
    if args.scaling == "strong":
        wm, wf, wt = args.replicas, rank * args.replicas, args.replicas * world
        pattern(False, wm, wf, wt)
        s.set_option("event_timing", 0)
        ww, _, _, _ = pattern(True, wm, wf, wt)
        s.set_option("event_timing", 1)
        if dist is not None:
            tw = torch.tensor(ww, dtype=torch.float64, device="cuda" if on_gpu_group else "cpu")
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            ww = tw.tolist()
        weak_value = wt * args.steps / statistics.median(ww)

    # ---- BASELINE configs[3] in every line: all 23 chromosomes at 500 kb x 20 replicas over the ranks (the sharding that scales) ----
    c4 = None
    if not args.no_side_figures and args.dtype == "f32":
        from chromosome3d_amd import batch
        c4 = config4_in_a_child() if (world == 1 and dist is None) else batch.bench_block(s, rank, world, dist, "cuda" if on_gpu_group else "cpu", sync_all)

    if rank == 0:
        value = total_replicas * args.steps / wall
        B = 4 * R + 72 * n                                             # algorithmic bytes per replica-step
        us_per_step_dev = 1e3 * dev_ms / args.steps
        launches_per_region = launches / reps
        bytes_per_launch = M * B * args.steps / max(launches_per_region, 1e-9)
        # the dominant kernel's own duration: start/end stamps of each multi-step launch (what a kernel trace shows);
        # on the per-step paths the event-bracketed region stands in (its launches run back to back)
        kernel_us_region = kern_us if kern_us > 0 else 1e3 * dev_ms
        avg_launch_us = kernel_us_region / max(launches_per_region, 1e-9)
        achieved = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9     # this rank's GPU: algorithmic bytes of a launch / its duration
        traffic, traffic_src = hbm_traffic_from_profiles(kernel, n, M)
        out = {
            "metric": f"SA-steps/sec (replica-steps/s, {total_replicas} replicas of chr1_500kb in all: {'+'.join(str(c) for c in per_rank)} per GPU, "
                      f"{args.scaling} scaling" + (f"; regions of {args.steps} steps inside the first {L_timed} of the schedule's {L}: MD and FIRE steps in front of the final minimisation"
                                                  if L_timed < L else "; regions over the whole schedule") + "); wall-clock per chromosome",
            "value": round(value, 1),
            "unit": "replica-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * wall / args.steps, 6),
            "higher_is_better": True,
            "scaling": args.scaling,
            # the weak-scaling figure under ONE key in every line, whatever --scaling says (strong: measured by a second pass of the same
            # call pattern with `--replicas` per GPU; weak: it IS `value`), so that lines of different rounds and modes stay comparable
            "weak_scaling_value": round(weak_value if weak_value is not None else value, 1),
            "weak_scaling_note": f"{args.replicas} replicas per GPU, same call pattern, same run",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "bundled Hi-C matrix chr1_500kb (tests/golden/inputs, exact float64 upper triangle); random-coil starts, seed 82364",
            "config": {"workload": f"{WORKLOAD}: N={n} beads, R={R} restraints, {total_replicas} replicas in all "
                                   f"({'+'.join(str(c) for c in per_rank)} per GPU), default schedule "
                                   f"(200 FIRE + 1000 hot MD + 972 cool MD + {MIN_STEPS} final minimisation [1000 two-point steps, then FIRE; no early exit here] = {L} steps; the timed regions lie in the first {L_timed})",
                       "replicas_per_gpu": per_rank, "parallelism": f"replica-sharded x{world}", "schedule_steps": L, "L_timed": L_timed,
                       "launch": {2: "one multi-step cluster launch per region", 0: "eager" if args.no_graph else "hipGraph",
                                  3: "fp64: one k64_step launch per step and replica group, hipGraph"}.get(path, "?")},
            "reps": reps,
            # two clocks per region, both maxima over the ranks: `value` uses the rank's own (from leaving the opening barrier to the return
            # of c3d_run_steps, which has waited for the solver's stream: the same interval as at one rank); the jointly bracketed one
            # (opening barrier -> closing device synchronise + barrier) adds the closing bracket's own latency
            "timing": {"value_uses": "max over ranks of each rank's clock: opening barrier left -> c3d_run_steps returned",
                       "barrier_to_barrier_ms": {"median": round(1e3 * bb_wall, 4), "min": round(1e3 * min(bb_walls), 4), "max": round(1e3 * max(bb_walls), 4)},
                       "value_barrier_to_barrier": round(total_replicas * args.steps / bb_wall, 1)},
            "region_wall_ms": {"median": round(1e3 * wall, 4), "min": round(1e3 * min(walls), 4), "max": round(1e3 * max(walls), 4)},
            "device_ms_per_region": round(dev_ms, 4),
            "us_per_step_device": round(us_per_step_dev, 4),
            "us_per_step_kernel": round(kernel_us_region / args.steps, 4),
            "process_totals": {"sa_steps": totals["sa_steps"], "step_kernel_launches": totals["launches"],
                               "note": "everything this process ran on the headline kernel (two full anneals in calls of --steps, the call pattern three times: "
                                       "untimed, timed, kernel-stamped; config 4 and the end-to-end run are child processes): a kernel trace's TotalDurationNs of "
                                       "the step kernel / sa_steps = us_per_step_kernel"},
            "graph_captures_in_timed_regions": int(captures_in_timed),
            "multi_step_launches_abandoned": int(fallbacks_in_timed),
            "wall_s_per_chromosome_full_schedule": round(full_wall, 5),
            "device_ms_full_schedule": round(full_dev_ms, 3),
            # every stage of the schedule, the final minimisation's two kernels included, in calls of --steps: the figure to compare rounds on
            # (rounds 1-4 sampled the final stage inside `value` at default arguments, rounds 5-6 do not: BASELINE.md section 5)
            "whole_schedule": {"value": round(total_replicas * L / full_wall, 1), "value_device": round(total_replicas * L / (1e-3 * full_dev_ms), 1),
                               "unit": "replica-steps/s", "steps": L, "calls_of": args.steps,
                               "note": "one complete fixed-length anneal (no exit test), wall = host clock around the calls, device = HIP-event pairs; max over ranks"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         # HBM bytes per launch from the PMC counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes, gfx950
                         # correction): read from the committed profile named here, NOT collected in this run; scaled from
                         # that profile's bytes per SA step to this run's SA steps per launch
                         "traffic": (round(traffic * args.steps / max(launches_per_region, 1e-9)) if traffic is not None else None),
                         "traffic_unit": "bytes per launch", "traffic_bytes_per_sa_step": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "avg_launch_us": round(avg_launch_us, 3),
                         "launches_per_region": round(launches_per_region, 2),
                         "algorithmic_bytes_per_launch": round(bytes_per_launch),
                         "algorithmic_bytes_per_sa_step": M * B,
                         "peak_measured": measured_stream_peaks(),
                         "note": "B = 4R + 72N per replica-step (SURVEY 8d) x replicas on this GPU; duration = the kernel's own start-to-end "
                                 "stamps (hipExtLaunchKernel events on the solver's stream, taken in a replay of the timed regions; median region / launches per region); the kernel is VALU-issue bound (DESIGN 5), the target "
                                 "matrix lives in registers / L2, so fabric traffic is below B by design"},
            # SURVEY 8d's secondary figure: the path is O(N^2) VALU work on L2-resident targets, so the low HBM fraction
            # is explained by the vector-ALU rate: algorithmic flops (30 R + 40 N per replica-step) against the fp32 vector peak
            "valu": {"flops_per_replica_step": 30 * R + 40 * n,
                     "achieved_tflops": round(M * (30 * R + 40 * n) / (kernel_us_region / args.steps * 1e-6) / 1e12, 2),
                     "peak_tflops": 157.3, "pairs_evaluated_per_replica_step": n * n,
                     "note": "every pair is evaluated from both of its rows (shipped potential: 16 packed + 6 scalar VALU instructions for two pair terms; "
                             "lanes without a column in the last, narrower block idle); peak = fp32 vector spec"},
            "collective": {"backend": (dist.get_backend() if dist is not None else None), "device": coll_dev, "world": world},
            "gather_ms": round(gather_ms, 3),
            "models_ranked": len(order),
            "spearman_if_invd_best_ranked": round(-float(allrec[order[0], 2]), 4),
            "spearman_if_invd_mean": round(-float(allrec[:, 2].mean()), 4),
            "spearman_reference_model": 0.8722,
            "e_noe_best": round(float(allrec[order[0], 1]), 1),
        }
        if c4 is not None:
            out["config4"] = c4
        if world == 1 and not args.no_side_figures and args.dtype == "f32":
            if path == 2:
                out["other_kernels"] = final_stage_in_a_child()
            out.update(side_figures(local_rank, IF, model, fire, stages, args, B))
            e2e = end_to_end(IF)
            if e2e is not None:
                out["end_to_end"] = e2e
        if not args.no_cpu_baseline and world == 1:          # the CPU leg runs on rank 0 of the one-GPU run only
            v, sample, all_cores = cpu_baseline(IF, d10, model, fire, stages)
            out["cpu_baseline"] = {"value": round(v, 1), "unit": "replica-steps/s", "cores": 1, "kind": "port",
                                   "sample": sample, "all_cores": all_cores,
                                   "reference_cpu_note": "the reference's CNS leg cannot run here (BASELINE.md 4); its Perl front half "
                                                         "took 3.3 s and its assessment 4.5 s per model at N=455 on one core (BASELINE.md 2)"}
        print(json.dumps(out), flush=True)
    s.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
