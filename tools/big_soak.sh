#!/bin/bash
# A longer run of every exactness soak on the build in the tree (about fifteen minutes on a GPU box); summary under gpurun_out/big_soak/summary.txt
cd "$(dirname "$0")/.."
O=gpurun_out/big_soak; mkdir -p $O
run() { local name=$1; shift; python3 "$@" > $O/$name.txt 2>&1; echo "== $* -> rc=$? $(tail -1 $O/$name.txt | cut -c1-360)" | tee -a $O/summary.txt; }
rm -f $O/summary.txt
run primlist tools/primlist_soak.py 5000 100000
run lastbounce tools/lastbounce_soak.py 3000 0
run many tools/many_soak.py 6000 200000
run many_large tools/many_soak.py 600 300000 large
run fuzz tools/fuzz_soak.py 5000 400000
run refcfg tools/ref_config_soak.py 3000 500000
run degenerate tools/degenerate_soak.py 1500 20000
run footprint tools/footprint_soak.py 150 30000
run chunk tools/chunk_soak.py 200 40000
