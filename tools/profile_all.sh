#!/bin/bash
# Round profile set on the GPU box: headline (xorwow, variant 6), counter-based generator (philox), the toleranced fast mode, config 4 grid
# kernel (pooled variant 13 closed/open, at the 256 spp BASELINE names, incl. WRITE/FETCH), config 5, per-config kernel times.  Summaries are made afterwards with tools/summarise_profile.py + update_roofline_json.py.
set -u
bash tools/profile.sh cfg2_xorwow && bash tools/profile.sh cfg2_philox --rng philox && bash tools/profile.sh cfg2_fast --fast && bash tools/profile.sh cfg3_xorwow --config cfg3 && \
bash tools/pmc_cfg4.sh cfg4_v13_closed closed 256 && bash tools/pmc_cfg4.sh cfg4_v13_open open 256 && \
bash tools/profile.sh cfg5_xorwow --config cfg5 --steps 200 && \
python3 tools/config_times.py > gpurun_out/config_times.log 2>&1
echo "profile_all rc=$?"
