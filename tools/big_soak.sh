#!/bin/bash
# A longer run of every exactness soak on the build in the tree (about fifteen minutes on a GPU box); summary under gpurun_out/big_soak/summary.txt
# Usage: tools/big_soak.sh [seed offset=0]  (added to every first seed: a second run on the same build covers new scenes)
cd "$(dirname "$0")/.."
O=gpurun_out/big_soak; mkdir -p $O
K=${1:-0}
run() { local name=$1; shift; python3 "$@" > $O/$name.txt 2>&1; echo "== $* -> rc=$? $(tail -1 $O/$name.txt | cut -c1-360)" | tee -a $O/summary.txt; }
rm -f $O/summary.txt
run primlist tools/primlist_soak.py 5000 $((100000+K))
run lastbounce tools/lastbounce_soak.py 3000 $((0+K))
run many tools/many_soak.py 6000 $((200000+K))
run many_large tools/many_soak.py 600 $((300000+K)) large
run fuzz tools/fuzz_soak.py 5000 $((400000+K))
run refcfg tools/ref_config_soak.py 3000 $((500000+K))
run degenerate tools/degenerate_soak.py 1500 $((20000+K))
run footprint tools/footprint_soak.py 150 $((30000+K))
run chunk tools/chunk_soak.py 200 $((40000+K))
