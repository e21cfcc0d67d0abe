#!/bin/bash
# AddressSanitizer + UBSan (+ float-cast-overflow) over the host-only translation unit of libc3d, then a ThreadSanitizer
# build of the same for the threaded matrix parser and concurrent callers of the host helpers (CPU builds; GPU
# sanitizers are not available on this pool).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
TMP=$(mktemp -d)
MATRIX="$ROOT/tests/golden/inputs/chr21_1mb_matrix.txt"; MODEL=$(ls "$ROOT"/tests/golden/models/chr21_1mb_rank07_a11.pdb)
g++ -std=c++17 -O1 -g -fsanitize=address,undefined,float-cast-overflow -fno-sanitize-recover=undefined,float-cast-overflow -fno-omit-frame-pointer \
    -o "$TMP/host_asan" "$HERE/host_asan_main.cpp" "$ROOT/chromosome3d_amd/csrc/c3d_host.cpp" -lpthread
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 "$TMP/host_asan" "$MATRIX" "$MODEL" "$TMP"
g++ -std=c++17 -O1 -g -fsanitize=thread -fno-omit-frame-pointer \
    -o "$TMP/host_tsan" "$HERE/host_asan_main.cpp" "$ROOT/chromosome3d_amd/csrc/c3d_host.cpp" -lpthread
TSAN_OPTIONS=halt_on_error=1 "$TMP/host_tsan" "$MATRIX" "$MODEL" "$TMP" tsan
rm -rf "$TMP"
