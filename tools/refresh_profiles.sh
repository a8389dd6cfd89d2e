#!/bin/bash
# After `gpurun -- bash tools/profile_all.sh` has brought gpurun_out/prof_* back: summarise into profiles/<round>/ and rebuild
# profiles/hbm_traffic.json + profiles/valu_roofline.json (fingerprinted; bench.py ignores them for any other build).
# Usage: tools/refresh_profiles.sh [round=r02]
set -e
cd "$(dirname "$0")/.."
R=${1:-r05}
python3 tools/summarise_profile.py cfg2_xorwow $R > /dev/null
python3 tools/summarise_profile.py cfg2_philox $R > /dev/null
python3 tools/summarise_profile.py cfg2_fast $R > /dev/null
python3 tools/summarise_profile.py cfg4_v13_closed $R > /dev/null
python3 tools/summarise_profile.py cfg4_v13_open $R > /dev/null
python3 tools/summarise_profile.py cfg5_xorwow $R > /dev/null
python3 tools/summarise_profile.py cfg3_xorwow $R > /dev/null
rm -f profiles/hbm_traffic.json profiles/valu_roofline.json
python3 tools/update_roofline_json.py cfg2_xorwow $R cfg2_xorwow_v6
python3 tools/update_roofline_json.py cfg2_philox $R cfg2_philox_v6
python3 tools/update_roofline_json.py cfg2_fast $R cfg2_xorwow_v100
python3 tools/update_roofline_json.py cfg3_xorwow $R cfg3_xorwow_v6
python3 tools/update_roofline_json.py cfg4_v13_closed $R cfg4_xorwow_v13      # counters of the 256-spp frame itself (tools/pmc_cfg4.sh)
python3 tools/update_roofline_json.py cfg4_v13_open $R cfg4open_xorwow_v13
python3 tools/update_roofline_json.py cfg5_xorwow $R cfg5_xorwow_v6
bash tools/isa.sh /tmp/pt_kernel_final.s
python3 tools/issue_model.py /tmp/pt_kernel_final.s pixel_kernelILi0ELi6ELb0ELi5E profiles/$R/cfg2_xorwow.json cfg2_xorwow_v6 | cut -c1-400
python3 tools/issue_model.py /tmp/pt_kernel_final.s pixel_kernelILi0ELi6ELb0ELi5E profiles/$R/cfg3_xorwow.json cfg3_xorwow_v6 | cut -c1-200
cp gpurun_out/config_times_vauto.json profiles/$R/config_times_auto.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import __graft_entry__ as ge
fp = ge.load_package().build_fingerprint()
rec = json.load(open("profiles/valu_roofline.json"))["cfg2_xorwow_v6"]
print("library", fp, "profile", rec["fingerprint"], "OK" if fp == rec["fingerprint"] else "STALE: rebuild / re-profile")
PY
