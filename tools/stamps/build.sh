#!/bin/bash
# Diagnostic library: libc3d with in-kernel cycle stamps (-DC3D_STAMPS).  Not part of the product build.
set -e
cd "$(dirname "$0")/../../chromosome3d_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -DC3D_STAMPS --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-kernarg-preload-count=16 \
    -shared -o ../../tools/stamps/libc3d_stamps.so c3d_device.hip c3d_embed.hip c3d_score.hip c3d_cluster.hip c3d_sym.hip c3d_f64.hip c3d_api.cpp c3d_host.cpp
