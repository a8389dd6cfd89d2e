#!/bin/bash
set -u
# The largest multi-rank worlds a ONE-GPU box allows (6 processes / 8 host threads sharing GPU 0), every gathered frame compared bit for bit with the
# 1-GPU frame; record: profiles/r05/mgpu_rehearsal.txt.  Usage (on a GPU box): bash tools/mgpu_rehearsal.sh
O=gpurun_out/rehearsal; mkdir -p $O
C="--steps 2 --warmup 1 --spp 16 --no-cpu-baseline --no-alt-rng --no-other-configs"
python3 bench.py --gpus 1 $C --dump $O/one.npy > $O/one.json 2> $O/one.err || echo "N=1 FAILED"
PT_BENCH_SHARED_GPU=1 PT_BENCH_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 6 $C --dump $O/dist6.npy > $O/dist6.json 2> $O/dist6.err || echo "dist6 FAILED"
PT_BENCH_SHARED_GPU=1 PT_FORCE_MGPU=1 timeout -k 10 300 python3 bench.py --gpus 8 --engine native $C --dump $O/nat8.npy > $O/nat8.json 2> $O/nat8.err || echo "nat8 FAILED"
PT_BENCH_SHARED_GPU=1 PT_FORCE_MGPU=1 timeout -k 10 300 python3 bench.py --gpus 8 --engine native --config cfg4 --steps 2 --warmup 1 --spp 16 --no-cpu-baseline --dump $O/nat8c4.npy > $O/nat8c4.json 2> $O/nat8c4.err || echo "nat8 cfg4 FAILED"
python3 bench.py --gpus 1 --config cfg4 --steps 2 --warmup 1 --spp 16 --no-cpu-baseline --dump $O/onec4.npy > $O/onec4.json 2> $O/onec4.err || echo "N=1 cfg4 FAILED"
python3 - <<'PY'
import numpy as np, json
O="gpurun_out/rehearsal/"
def same(a,b):
    try: return bool(np.array_equal(np.load(O+a).view(np.uint32), np.load(O+b).view(np.uint32)))
    except Exception as e: return repr(e)
def line(f):
    try:
        j=json.loads([l for l in open(O+f) if l.startswith('{"metric"')][-1]); return {k:j.get(k) for k in ("n_gpus","ms_per_step","value")}, j["config"].get("tiling"), j["config"].get("engine")
    except Exception as e: return repr(e)
print("dist 6 ranks sharing GPU 0 (gloo):", line("dist6.json"), "frame == 1-GPU frame:", same("dist6.npy","one.npy"))
print("native 8 ranks sharing GPU 0     :", line("nat8.json"), "frame == 1-GPU frame:", same("nat8.npy","one.npy"))
print("native 8 ranks, cfg4 (16 spp)    :", line("nat8c4.json"), "frame == 1-GPU frame:", same("nat8c4.npy","onec4.npy"))
PY
rm -f $O/*.npy
