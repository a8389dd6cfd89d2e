#!/bin/bash
# bench.py for every BASELINE configuration, one JSON line each under gpurun_out/bench_<cfg>.json (copied into profiles/<round>/ afterwards)
python bench.py > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; for c in cfg3 cfg4 cfg4open cfg5; do python bench.py --config $c --no-other-configs > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; done; ls -la gpurun_out/bench_cfg*.json
