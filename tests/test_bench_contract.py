"""bench.py's contract (the driver runs it with --steps 20 --warmup 5): one JSON line, the keys the brief names, no graph capture
or abandoned launch inside the timed regions, parity keys from a full anneal, both scaling modes over two ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline")


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_driver_arguments_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--reps", "10", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "replica-steps/s" and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert "chr1_500kb" in d["config"]["workload"] and d["config"]["replicas_per_gpu"] == [20]
    assert "20 replicas of chr1_500kb in all: 20 per GPU" in d["metric"]
    # both clocks of a region are in the line; at one rank they differ by the loop's bookkeeping only
    t = d["timing"]
    assert t["barrier_to_barrier_ms"]["median"] >= d["region_wall_ms"]["median"] and t["value_barrier_to_barrier"] <= d["value"] * 1.001
    assert t["value_barrier_to_barrier"] > 0.9 * d["value"]
    assert d["graph_captures_in_timed_regions"] == 0 and d["multi_step_launches_abandoned"] == 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # achieved = algorithmic bytes of a launch / the kernel's own duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 0.01 * r["achieved"]
    assert r["algorithmic_bytes_per_sa_step"] == 20 * (4 * 101426 + 72 * 455)
    assert r["traffic"] is not None and r["traffic_source"].startswith("profiles/")        # from a committed counter profile
    assert "k_cluster" in r["kernel"]
    # whole-job throughput: consistent with its own wall clock, and in the range this machine delivers
    assert abs(d["value"] - 20 * 20 / (d["region_wall_ms"]["median"] * 1e-3)) < 0.01 * d["value"]
    assert 2.0e6 < d["value"] < 1.0e7
    # the line always carries parity: best-ranked replica of a full anneal against the bundled model (north star: +-0.01)
    assert abs(d["spearman_if_invd_best_ranked"] - d["spearman_reference_model"]) <= 0.01
    assert d["models_ranked"] == 20


@pytest.mark.gpu
@pytest.mark.parametrize("mode,counts", [("strong", [10, 10]), ("weak", [20, 20])])
def test_two_ranks_rehearsal(mode, counts):
    """Two ranks (gloo rendezvous; both use this box's one GPU, so the value says nothing about scaling): replica split,
    max-over-ranks timing and the gather run as they will under RCCL."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, C3D_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--reps", "5",
                        "--scaling", mode, "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == mode and d["config"]["replicas_per_gpu"] == counts
    assert d["models_ranked"] == sum(counts)
    assert f"{sum(counts)} replicas of chr1_500kb in all: {counts[0]}+{counts[1]} per GPU, {mode} scaling" in d["metric"]
    # BASELINE configs[3] rides in every multi-rank line: 23 chromosomes x 20 replicas, LPT over the ranks, one gather
    c4 = d["config4"]
    assert c4["chromosomes_ranked"] == 23 and len(c4["per_rank"]) == 2 and sum(r["chromosomes"] for r in c4["per_rank"]) == 23
    assert c4["standins"] == ["chr2_500kb"] and c4["wall_s"] >= max(r["solve_s"] for r in c4["per_rank"]) > 0
    ra, rb = (r["restraints"] for r in c4["per_rank"])
    assert abs(ra - rb) <= 0.15 * (ra + rb)                              # LPT: the two ranks carry about the same restraint count
    assert d["timing"]["barrier_to_barrier_ms"]["median"] >= d["region_wall_ms"]["median"]
    # the weak figure rides under one key in every line; in weak mode it is the value itself
    assert "weak_scaling_value" in d and (mode == "strong" or d["weak_scaling_value"] == d["value"])
    assert abs(d["spearman_if_invd_best_ranked"] - d["spearman_reference_model"]) <= 0.01


# ---- how bench.py / the batch driver get their ranks (chromosome3d_amd/launch.py) --------------------------------------
def test_gpus_flag_fails_loudly_without_that_many_gpus():
    """`python bench.py --gpus 2` on a machine with fewer than 2 GPUs (this container: none; the one-GPU box: one) stops with a
    message naming the shortfall instead of running one rank and calling it two.  Runs before anything touches HIP."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "C3D_BENCH_BACKEND")}
    for cmd in ([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], ["-m", "chromosome3d_amd.batch", "--gpus", "2"]):
        p = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
        import torch
        if torch.cuda.device_count() >= 2:
            pytest.skip("this machine has two GPUs")
        assert p.returncode != 0 and "--gpus 2" in p.stderr and "GPU(s)" in p.stderr, (p.returncode, p.stderr[-500:])
        assert '"metric"' not in p.stdout


def test_world_size_must_match_the_gpus_flag():
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK")}, WORLD_SIZE="4", RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


def test_launcher_rank_bookkeeping(monkeypatch):
    from chromosome3d_amd import launch
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert launch.ensure_ranks(1, []) == (0, 0, 1)
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "5")
    assert launch.ensure_ranks(8, []) == (5, 5, 8)
    with pytest.raises(SystemExit):
        launch.ensure_ranks(4, [])


@pytest.mark.gpu
def test_bench_through_rccl_at_one_rank():
    """`--dist`: init_process_group("nccl"), the all_reduce(MAX) of the region times on the device and the device-side
    all_gather of the model records all execute on this box's one MI355X (the N > 1 case is the same code over more ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "C3D_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dist", "--steps", "20", "--warmup", "5", "--reps", "5",
                        "--no-cpu-baseline", "--no-side-figures"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["config"]["replicas_per_gpu"] == [20] and d["models_ranked"] == 20
    assert d["collective"] == {"backend": "nccl", "device": "cuda", "world": 1}
    assert 2.0e6 < d["value"] < 1.0e7


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts both ranks (children under torch.distributed.run) and
    relays rank 0's line.  gloo rehearsal: this box has one GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["C3D_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--reps", "5",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    # no --scaling given: more than one rank defaults to the north star's case, 20 replicas IN ALL; the weak figure rides along
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["replicas_per_gpu"] == [10, 10] and d["collective"]["backend"] == "gloo"
    assert "weak_scaling_value" in d and "config4" in d and d["config4"]["wall_s"] > 0


@pytest.mark.gpu
def test_side_figures_ride_in_the_one_gpu_line():
    """The driver's line (one GPU) also carries the precision-matched fp64 leg and the step rates of configs 2 and 5."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--reps", "5", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert 1.2e6 < d["value_f64"] < d["value"] and 0.05 < d["frac_f64"] < 0.3 and "k64_step" in d["f64"]["kernel"]
    assert d["value"] / d["value_f64"] < 3.5                             # round 3: 6.7 x (a naive twin kernel); the fp64 vector rate is half of fp32's
    c2, c5 = d["other_configs"]["config2"], d["other_configs"]["config5"]
    assert c2["n"] == 37 and c2["replicas"] == 20 and 0.5 < c2["us_per_step_device"] < 5.0
    assert c5["n"] == 2500 and c5["replicas"] == 8 and c5["restraints"] == 3113760 and 10.0 < c5["us_per_step_device"] < 60.0
    r = d["roofline"]
    assert r["traffic_unit"] == "bytes per launch" and r["traffic"] == round(r["traffic_bytes_per_sa_step"] * 20)
    # config 4 on this one GPU and the user-facing path of chromosome3D.pl for the headline matrix as a child process
    assert d["config4"]["chromosomes_ranked"] == 23 and 0.1 < d["config4"]["wall_s"] < 10.0
    e = d["end_to_end"]
    assert 0.01 < e["job_s"] < 10.0 and e["process_wall_s"] >= e["job_s"] and abs(sum(e["phases_s"].values()) - e["job_s"]) < 0.05


# ---- rehearsal of what the first real SCALE run executes, at more than two ranks -----------------------------
# A GPU box allows at most 6 processes on its card: FOUR ranks run here on the one GPU (gloo rendezvous); the EIGHT-rank line — 20 replicas
# as 3,3,3,3,2,2,2,2, 23 chromosomes by LPT over 8 ranks, the 8-way gather — runs on the CPU with a stand-in solver
# (tests/test_sharding_gloo.py::test_bench_line_at_eight_ranks).
@pytest.mark.gpu
def test_four_ranks_started_by_bench_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["C3D_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5", "--reps", "3",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=1100)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["config"]["replicas_per_gpu"] == [5, 5, 5, 5] and d["models_ranked"] == 20
    assert "20 replicas of chr1_500kb in all: 5+5+5+5 per GPU, strong scaling" in d["metric"]
    assert d["collective"] == {"backend": "gloo", "device": "cpu", "world": 4} and d["weak_scaling_value"] > 0
    c4 = d["config4"]
    assert c4["chromosomes_ranked"] == 23 and len(c4["per_rank"]) == 4 and sum(r["chromosomes"] for r in c4["per_rank"]) == 23
    assert all(r["chromosomes"] >= 4 for r in c4["per_rank"])
    loads = [r["restraints"] for r in c4["per_rank"]]
    assert max(loads) <= 1.15 * (sum(loads) / 4)                         # LPT: no rank carries 15 % more than its share
    assert abs(d["spearman_if_invd_best_ranked"] - d["spearman_reference_model"]) <= 0.01


@pytest.mark.gpu
def test_batch_driver_at_four_ranks_prints_the_one_rank_results_on_stdout():
    """`python -m chromosome3d_amd.batch --gpus 4` (the module starts its ranks itself; gloo on this box's one GPU): the per-chromosome
    ranking and truncated energies equal the one-rank run's, and WITHOUT --json the result table arrives on STDOUT like the one-rank
    path's (round 4's relay sent every line that did not start with '{' to stderr: `> out.txt` was empty)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "C3D_BENCH_BACKEND")}
    one = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--json"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    env4 = dict(env, C3D_BENCH_BACKEND="gloo")
    four = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--gpus", "4", "--json"], cwd=ROOT, env=env4, capture_output=True, text=True, timeout=900)
    assert four.returncode == 0, four.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in four.stdout.splitlines() if l.startswith("{")][-1])
    assert a["world"] == 1 and b["world"] == 4 and len(b["chromosomes"]) == 23 and a["chromosomes"] == b["chromosomes"]
    table = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--gpus", "2", "--pattern", "chr2"], cwd=ROOT, env=env4, capture_output=True,
                           text=True, timeout=900)
    assert table.returncode == 0, table.stderr[-2000:]
    rows = [l for l in table.stdout.splitlines() if l.lstrip().startswith("chr2")]
    assert "chromosomes x 20 replicas on 2 rank(s)" in table.stdout and len(rows) >= 3, table.stdout[-1500:]      # chr2, chr20..chr23 at 500 kb
    assert "[Gloo]" not in table.stdout


def test_launcher_relays_results_on_stdout_and_only_chatter_on_stderr():
    from chromosome3d_amd import launch
    chatter = ["[Gloo] Rank 0 is connected to 7 peer ranks. Expected number of connected peer ranks is : 7\n",
               "[W1005 03:52:11.123456789 ProcessGroupGloo.cpp:75] Warning: something\n", "W1005 03:52:11.000000 123 torch/distributed/run.py:793] x\n",
               "[rank3]:[W1005 03:52:11.1 ProcessGroupNCCL.cpp:4] y\n", "box:123:456 [0] NCCL INFO Bootstrap : Using lo\n"]
    ours = ['{"metric": "SA-steps/sec"}\n', "23 chromosomes x 20 replicas on 8 rank(s); rank 0 solved 3 of them in 0.11 s\n",
            "  chr1_500kb   N= 455 models=20 best: replica  7 E_noe=   4370445.0 Spearman(IF,1/d)=0.8738  anneal 12.9 ms\n",
            "  total wall incl. load/score 0.52 s; anneal device time summed over chromosomes 210.0 ms\n", "\n", "Warning: not a log record\n"]
    assert all(launch.is_chatter(l) for l in chatter) and not any(launch.is_chatter(l) for l in ours)
