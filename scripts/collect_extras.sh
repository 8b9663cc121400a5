#!/bin/bash
# The round's remaining evidence, on ONE box in ONE call, after scripts/collect_profiles.sh <tag> (same call if the limit allows):
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r03 && bash scripts/collect_extras.sh r03'
# -> gpurun_out/extras_<tag>/ : stencils.txt, shard_ax_lab.txt, cg_schedules.txt, band_sweep.txt, shard_k20.json, shard_k500.json and
#    the kernel trace of the one-rank sharded rehearsal (scripts/shard_trace.sh); copy the text files to profiles/<tag>_*.
set -e -o pipefail
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/extras_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
for a in "200 200 200 27 1" "256 256 128 27 1" "128 128 488 27 1" "192 192 217 27 1" "1000 1000 8 27 1" "200 200 200 7 1" "128 128 488 7 1" \
         "1000 1000 8 7 1" "100 100 100 27 3" "128 128 128 7 3" "100 100 100 7 4" "140 140 140 7 2" "100 100 100 27 2" "1500 1500 1 27 3"; do
    timeout -k 10 120 python3 scripts/stencil27.py $a 2>/dev/null >> "$OUT/stencils.txt"
done
echo "stencils done"
timeout -k 10 200 python3 scripts/shard_ax_lab.py > "$OUT/shard_ax_lab.txt" 2>/dev/null
timeout -k 10 200 python3 scripts/cg_schedules.py > "$OUT/cg_schedules.txt" 2>/dev/null
LCG_HIP_DEBUG_BINNED=1 timeout -k 10 200 python3 scripts/band_sweep.py 2>&1 | grep -E "^band|tiled choice" > "$OUT/band_sweep.txt"
echo "labs done"
export LCG_HIP_FORCE_COMM=1 LCG_HIP_DIST_MODE=2 MASTER_PORT=29563
timeout -k 10 200 python3 bench.py --rows 1250000 --steps 20 --warmup 5 --no-cpu-baseline --no-variants > "$OUT/shard_k20.json" 2> "$OUT/shard_k20.err"
timeout -k 10 200 python3 bench.py --rows 1250000 --steps 500 --warmup 5 --no-cpu-baseline --no-variants > "$OUT/shard_k500.json" 2> "$OUT/shard_k500.err"
unset LCG_HIP_FORCE_COMM LCG_HIP_DIST_MODE MASTER_PORT
echo "sharded rehearsal done"
bash scripts/shard_trace.sh $TAG > "$OUT/shard_trace.txt" 2>&1
echo "shard trace done"
