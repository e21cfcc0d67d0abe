#!/bin/bash
# AddressSanitizer + UBSan over the host-only translation unit of libc3d (CPU build).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
TMP=$(mktemp -d)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o "$TMP/host_asan" "$HERE/host_asan_main.cpp" "$ROOT/chromosome3d_amd/csrc/c3d_host.cpp"
ASAN_OPTIONS=detect_leaks=1 "$TMP/host_asan" "$ROOT/tests/golden/inputs/chr21_1mb_matrix.txt" "$ROOT"/tests/golden/models/chr21_1mb_rank07_a11.pdb "$TMP"
rm -rf "$TMP"
