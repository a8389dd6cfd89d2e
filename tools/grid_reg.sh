#!/bin/bash
# Round 5's ball-reach registration of the grid builder (csrc/pt_grid.h, PT_GRID_EXACT_REG): A/B against the bounding-box rule and other
# cell densities (builds from tools/build_alt.sh under cuda-pathtrace_amd/alt), the mutant self-test (a ball 7 % too small must be caught
# by the many-sphere soak) and the soaks on the build in the tree.  Record: gpurun_out/grid_reg.txt -> profiles/r05/grid_reg.txt
cd "$(dirname "$0")/.."
O=gpurun_out/grid_reg.txt; : > $O
echo "## A/B, config 4 at 256 spp (noreg = -DPT_GRID_EXACT_REG=0; reg_cX = cells per sphere X, main = 2.75)" >> $O
timeout -k 10 400 python3 tools/cfg4_ab.py 256 main noreg reg_c2.0 reg_c2.4 reg_c3.0 main noreg >> $O 2>&1 || exit 1
echo "## mutant (-DPT_GRID_REG_MUTANT=1: reach x 0.93): many_soak.py 200 0 must report differences" >> $O
PT_LIB_OVERRIDE=$PWD/cuda-pathtrace_amd/alt/regmut/libptcore.so timeout -k 10 300 python3 tools/many_soak.py 200 0 2>&1 | tail -1 | cut -c1-300 >> $O
echo "## the build in the tree" >> $O
for c in "many_soak.py 1500 700000" "many_soak.py 150 710000 large" "fuzz_soak.py 1500 720000" "degenerate_soak.py 400 30000"; do
  echo "== $c" >> $O; timeout -k 10 500 python3 tools/$c 2>&1 | tail -1 | cut -c1-300 >> $O || exit 1
done
tail -12 $O
