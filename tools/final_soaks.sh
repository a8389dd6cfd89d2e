#!/bin/bash
# The exactness soaks and the whole GPU suite on the build in the tree, one after the other; results under gpurun_out/final_soaks/
# (copy the summary lines into profiles/<round>/final_soaks.txt).  About ten minutes on a GPU box.  Usage: tools/final_soaks.sh
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/final_soaks
mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest -m gpu rc=$? $(tail -1 $O/pytest_gpu.txt)" | tee $O/summary.txt
run() { local name=$1; shift; echo "== $*" >> $O/summary.txt; python3 "$@" > $O/$name.txt 2>&1; echo "rc=$? $(tail -1 $O/$name.txt | cut -c1-400)" | tee -a $O/summary.txt; }
run primlist tools/primlist_soak.py 1500 5000
run lastbounce tools/lastbounce_soak.py 600 50000
run many tools/many_soak.py 1200 30000
run many_large tools/many_soak.py 150 40000 large
run fuzz tools/fuzz_soak.py 1500 100000
run refcfg tools/ref_config_soak.py 1500 60000
run footprint tools/footprint_soak.py 100 10000
run chunk tools/chunk_soak.py 100 8000
run degenerate tools/degenerate_soak.py 600 4000
run grid_check tools/grid_check.py 8
cp gpurun_out/grid_check.json $O/grid_check.json 2>/dev/null
echo "fingerprint $(python3 -c "import __graft_entry__ as g; print(g.load_package().build_fingerprint())")" | tee -a $O/summary.txt
