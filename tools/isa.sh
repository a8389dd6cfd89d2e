#!/bin/bash
# Emit the gfx950 ISA of pt_kernel.hip to $1 (default /tmp/pt_kernel.s) with the library's flags.
out=${1:-/tmp/pt_kernel.s}
cd "$(dirname "$0")/../cuda-pathtrace_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-bitwise-instead-of-logical --cuda-device-only -S pt_kernel.hip -o "$out" 2>&1 | grep -v hip-link
exit 0
