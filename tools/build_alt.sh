#!/bin/bash
# build_alt.sh <name> <extra hipcc flags...>: builds cuda-pathtrace_amd/alt/<name>/libptcore.so for A/B experiments
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p cuda-pathtrace_amd/alt/$N
cd cuda-pathtrace_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -shared -o ../alt/$N/libptcore.so pt_kernel.hip pt_fast.hip pt_capi.hip pt_mgpu.hip pt_display.hip pt_host.cpp -ldl -lpthread
