set -e
mkdir -p gpurun_out/r05b
for cm in 1 2 4 8; do for hm in 1 2; do
python tools/parity_sweep.py "{\"quiet\":1,\"cool_mult\":$cm,\"hot_mult\":$hm}" > gpurun_out/r05b/sweep_c${cm}_h${hm}.txt 2>&1
done; done
python tools/parity_sweep.py '{"cool_mult":4}' > gpurun_out/r05b/sweep_c4_full.md 2>&1
python tools/parity_sweep.py '{"cool_mult":8,"hot_mult":2}' > gpurun_out/r05b/sweep_c8h2_full.md 2>&1
