#!/bin/bash
# Profiles of the round, on the GPU box (outputs under gpurun_out/<tag>/, to be copied into profiles/):
#   kernel traces of the two bench commands (default and the driver's --steps 20 --warmup 5) and the HBM-traffic counters of the
#   headline kernel (separate --pmc passes, as MI355X_MICROARCH.md prescribes).    bash tools/profile_round.sh r02 [traces]
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/trace_default -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_default_under_rocprof.json 2> $O/trace_default.log
rocprofv3 --kernel-trace --stats -d $O/trace_s20 -o t --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_s20_under_rocprof.json 2> $O/trace_s20.log
if [ "$2" != "traces" ]; then
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p --output-format csv -- python3 $R/tools/pmc_run_cluster.py 20 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p --output-format csv -- python3 $R/tools/pmc_run_cluster.py 20 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o p --output-format csv -- python3 $R/tools/pmc_run_cluster.py 20 > $O/pmc_sq.log 2>&1
cd $R
python tools/pmc_summarise.py $O/pmc_fetch $O/pmc_write k_cluster --json $O/hbm_traffic_k_cluster.json --n 455 --replicas 20 --steps-per-dispatch 2000 > $O/pmc_hbm_summary.txt
python tools/pmc_summarise.py $O/pmc_sq k_cluster > $O/pmc_sq_summary.txt
fi
cd /tmp
# the fp64 step kernel (option precision = 64) and config 5's per-step kernel: kernel traces; config 5's SQ counters
rocprofv3 --kernel-trace --stats -d $O/trace_f64 -o t --output-format csv -- python3 $R/bench.py --dtype f64 --steps 200 --warmup 400 --no-cpu-baseline --no-side-figures > $O/bench_f64_under_rocprof.json 2> $O/trace_f64.log
if [ "$2" != "traces" ]; then
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq_config5 -o p --output-format csv -- python3 $R/tools/config5_profile_run.py 0 > $O/pmc_sq_config5.log 2>&1
python3 $R/tools/pmc_summarise.py $O/pmc_sq_config5 k_step > $O/pmc_sq_config5_summary.txt
# config 5's HBM traffic per step (round 6: the one HBM-streaming configuration; its step time has moved between boxes, 25.4-33 us):
# a dispatch of the wide per-step kernel = one SA step of a replica group (4 of the 8 replicas)
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_config5 -o p --output-format csv -- python3 $R/tools/config5_profile_run.py 0 > $O/pmc_fetch_config5.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_config5 -o p --output-format csv -- python3 $R/tools/config5_profile_run.py 0 > $O/pmc_write_config5.log 2>&1
python3 $R/tools/pmc_summarise.py $O/pmc_fetch_config5 $O/pmc_write_config5 k_step --json $O/hbm_traffic_config5_k_step.json --n 2500 --replicas 4 --steps-per-dispatch 1 > $O/pmc_hbm_config5_summary.txt
(rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -8) > $O/clocks_after_config5.txt 2>&1 || true
fi
cd $R
python tools/trace_check.py $O/trace_default/t_kernel_stats.csv $O/bench_default_under_rocprof.json $O/trace_s20/t_kernel_stats.csv $O/bench_s20_under_rocprof.json > $O/trace_vs_bench.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_s20.json 2> $O/bench_s20.err
echo done
