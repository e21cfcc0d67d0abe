#!/bin/bash
# run_all_amd.sh — the reference's test.sh (test.sh:4-12: every chromosome at 1 Mb and 500 kb, one background
# chromosome3D.pl per matrix) for the MI355X build: ONE c3d_batch process (chromosome3d_amd/csrc/c3d_batch_main.cpp),
# one host thread and one libc3d context per GPU, matrices dealt to the GPUs largest-first by N^2.
#   bin/run_all_amd.sh <input dir with *_matrix.txt> <output root> [n_gpus=1] [c3d_batch options: -m 20 -k 11 -a 0.5 ...]
# (per-matrix Perl driver with the reference's own command line: bin/chromosome3D_amd.pl)
set -u
IN=${1:?input directory}; OUT=${2:?output root}; NG=${3:-1}; shift; shift; shift || true
HERE=$(cd "$(dirname "$0")" && pwd)
exec "$HERE/../chromosome3d_amd/_lib/c3d_batch" "$IN" --out "$OUT" --devices "$NG" "$@"
