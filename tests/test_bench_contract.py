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
    assert ("weak_scaling_value" in d) == (mode == "strong")
    assert abs(d["spearman_if_invd_best_ranked"] - d["spearman_reference_model"]) <= 0.01
