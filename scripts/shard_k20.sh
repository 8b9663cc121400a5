#!/bin/bash
# The per-solve fixed cost of the sharded loop: one-rank rehearsal at the 8-way shard size (1.25M rows), K = 20
# against K = 500, alternating, direct paths forced; then a kernel trace of the K = 20 run (durations and the gaps between kernels of
# ONE solve: scripts/trace_gaps.py).    gpurun -- 'bash scripts/shard_k20.sh r04a > gpurun_out/shard_k20_r04a.txt 2>&1'
set -o pipefail
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
export LCG_HIP_FORCE_COMM=1 LCG_HIP_DIST_MODE=2 MASTER_PORT=29565
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$1', round(d['ms_per_step']*1e3,1), 'us/iteration (median of', d['timed_repetitions'], '; best', round(1e6/d['value_max'],1), ')', d.get('comm_probe',{}).get('chosen',''))"; }
for i in 1 2 3; do
  for k in 20 500; do
    timeout -k 10 200 python3 bench.py --rows 1250000 --steps $k --warmup 5 --reps 7 --no-cpu-baseline --no-variants 2>/dev/null | line $k
  done
done
OUT=$REPO/gpurun_out/shard_k20_trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o prof -- \
    python3 "$REPO/bench.py" --rows 1250000 --steps 20 --warmup 5 --reps 3 --no-cpu-baseline --no-variants > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "trace rc $?"
ls -la "$OUT"/*/ 2>/dev/null | head
