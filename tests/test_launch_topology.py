"""chromosome3d_amd.launch counts GPUs from the KFD topology in sysfs (no HIP, no torch in the launcher parent)."""
import os

from chromosome3d_amd import launch


def _node(root, k, simd):
    d = os.path.join(root, str(k))
    os.makedirs(d)
    with open(os.path.join(d, "properties"), "w") as fh:
        fh.write(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")


def test_visible_gpus_counts_kfd_nodes_with_simds(tmp_path, monkeypatch):
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    root = str(tmp_path / "nodes")
    os.makedirs(root)
    _node(root, 0, 0)                      # the CPU node
    _node(root, 1, 0)
    for k in range(2, 10):
        _node(root, k, 1024)               # eight GPUs
    assert launch.visible_gpus(root) == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3")
    assert launch.visible_gpus(root) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "5")
    assert launch.visible_gpus(root) == 1
    # no topology (this container, a sandbox): the launcher lets the ranks find out
    assert launch.visible_gpus(str(tmp_path / "absent")) is None


def test_launcher_parent_does_not_import_torch():
    """The parent that starts the ranks must not load torch (and with it HIP): counted in a fresh interpreter."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; from chromosome3d_amd import launch; launch.visible_gpus(); launch.free_port(); print('torch' in sys.modules)"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "False", out.stdout + out.stderr
