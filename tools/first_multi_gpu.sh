#!/bin/bash
# First contact with a multi-GPU node (none was available to any round so far): everything that has only ever run with ranks
# sharing one device, now between distinct devices over xGMI -- and BASELINE.md section 4's table filled from it.
#   1. tests/test_mgpu_gpu.py (the distinct-device RCCL gather, the band policy between devices) and the bench multi-rank tests;
#   2. bench.py --gpus {1,2,4,8} for cfg2 and cfg3 -- and the 1000-sphere frames cfg4 / cfg4open, whose row tiles are uneven
#      (predicted x2.4 / x2.2 at N = 8 from one-GPU tile times) -- with both engines (one process per GPU + torch.distributed/RCCL; one process,
#      pt_mgpu_* with banded exchange), every gathered frame compared bit for bit with the 1-GPU frame;
#   3. the table: kernel ms, ms per frame (pipelined), one frame's latency, exposed gather, Msamples/s, efficiency.
# Nothing here touches a GPU in this shell: every step is a child process.  Usage: tools/first_multi_gpu.sh [outdir=gpurun_out/mgpu]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/mgpu}
mkdir -p "$OUT"
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
echo "devices: $NGPU" | tee "$OUT/summary.txt"
python3 tools/tile_skew.py 2 > "$OUT/tile_skew.txt" 2>&1  # the one-GPU prediction from this box, to set beside the runs
python3 -m pytest tests/test_mgpu_gpu.py tests/test_bench_multirank_gpu.py -q -x > "$OUT/tests.log" 2>&1; echo "tests rc=$? ($(tail -1 "$OUT/tests.log"))" | tee -a "$OUT/summary.txt"
for CFG in cfg2 cfg3 cfg4 cfg4open; do
  python3 bench.py --gpus 1 --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-alt-rng --no-other-configs --dump "$OUT/${CFG}_n1.npy" > "$OUT/${CFG}_n1_dist.json" 2> "$OUT/${CFG}_n1_dist.err" || echo "$CFG N=1 FAILED" | tee -a "$OUT/summary.txt"
  for N in 2 4 8; do
    [ "$N" -le "$NGPU" ] || continue
    for ENGINE in dist native; do
      python3 bench.py --gpus $N --engine $ENGINE --config $CFG --steps 5 --warmup 1 --no-cpu-baseline --no-alt-rng --dump "$OUT/${CFG}_n${N}_${ENGINE}.npy" \
        > "$OUT/${CFG}_n${N}_${ENGINE}.json" 2> "$OUT/${CFG}_n${N}_${ENGINE}.err" || echo "$CFG N=$N $ENGINE FAILED (see $OUT/${CFG}_n${N}_${ENGINE}.err)" | tee -a "$OUT/summary.txt"
    done
  done
done
python3 - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import glob, json, os, sys
import numpy as np
out = sys.argv[1]
print("| config | N | engine | frame = 1-GPU frame | kernel ms (slowest rank) | ms/frame pipelined | one frame latency ms | exposed gather ms | Msamples/s | efficiency |")
print("|---|---|---|---|---|---|---|---|---|---|")
for cfg in ("cfg2", "cfg3", "cfg4", "cfg4open"):
    base, ref = None, None
    p1 = os.path.join(out, f"{cfg}_n1.npy")
    if os.path.exists(p1):
        ref = np.load(p1, mmap_mode="r")
    for f in sorted(glob.glob(os.path.join(out, f"{cfg}_n*_*.json")), key=lambda s: (int(s.split("_n")[1].split("_")[0]), s)):
        lines = [l for l in open(f) if l.startswith('{"metric"')]
        if not lines:
            print(f"| {cfg} | {os.path.basename(f)} | - | NO RESULT | | | | | | |")
            continue
        j = json.loads(lines[-1])
        n, eng = j["n_gpus"], j["config"]["engine"]
        if n == 1:
            base = j["value"]
        dump = f.replace(".json", ".npy") if n > 1 else p1
        same = "-"
        if ref is not None and os.path.exists(dump):
            same = "yes" if np.array_equal(np.load(dump, mmap_mode="r").view(np.uint32), ref.view(np.uint32)) else "NO"
        eff = f"{j['value'] / (base * n) * 100:.1f} %" if base else "-"
        print(f"| {cfg} | {n} | {eng} | {same} | {j['roofline']['kernel_ms']} | {j['ms_per_step']} | {j.get('frame_latency_ms', '-')} | "
              f"{j.get('exchange_exposed_ms', '-')} | {j['value']} | {eff} |")
PY
rm -f "$OUT"/*.npy  # (the frames: 59 MB for cfg2, 940 MB for cfg3 each)
echo "table: $OUT/summary.txt (paste into BASELINE.md section 4 and DESIGN.md section 5)"
