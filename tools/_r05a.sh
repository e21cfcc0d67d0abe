set -e
mkdir -p gpurun_out/r05a
python tools/parity_sweep.py > gpurun_out/r05a/parity_sweep_all45.md 2> gpurun_out/r05a/parity.err
python tools/cross_resolution.py > gpurun_out/r05a/cross_resolution.md 2> gpurun_out/r05a/cross.err
python tools/fire_convergence.py > gpurun_out/r05a/fire_convergence.txt 2>&1 || true
python -m pytest tests/test_gpu_parity.py -q -x -k "fp32_product" -s > gpurun_out/r05a/t_fp.log 2>&1 || true
python bench.py --steps 20 --warmup 5 > gpurun_out/r05a/bench20.json 2> gpurun_out/r05a/bench20.err
python bench.py > gpurun_out/r05a/bench_default.json 2> gpurun_out/r05a/bench_default.err
tail -3 gpurun_out/r05a/parity_sweep_all45.md | cut -c1-1500
