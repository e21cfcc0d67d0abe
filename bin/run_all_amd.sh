#!/bin/bash
# run_all_amd.sh — the reference's test.sh (test.sh:4-12: every chromosome at 1 Mb and 500 kb in the
# background) for the MI355X driver: one job per matrix, round-robin over the GPUs of the node, at most
# one job per GPU at a time (a 20-replica chromosome takes ~0.3 s end to end, so jobs are queued per GPU
# rather than oversubscribed).
#   bin/run_all_amd.sh <input dir with *_matrix.txt> <output root> [n_gpus=1] [extra driver options]
set -u
IN=${1:?input directory}; OUT=${2:?output root}; NG=${3:-1}; shift; shift; shift || true
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$OUT"
g=0
for gpu in $(seq 0 $((NG - 1))); do
  (
    i=0
    for m in $(ls "$IN"/*_matrix.txt | sort -V); do
      if [ $((i % NG)) -eq "$gpu" ]; then
        id=$(basename "$m" _matrix.txt)
        echo "Running job for ${id} on GPU ${gpu}.."
        perl "$HERE/chromosome3D_amd.pl" -if "$m" -o "$OUT/$id" --device "$gpu" "$@" &> "$OUT/$id.log" || echo "FAILED: $id (see $OUT/$id.log)"
      fi
      i=$((i + 1))
    done
  ) &
done
wait
echo "all jobs finished; models under $OUT/<chromosome>/<ID>_model1..5.pdb"
