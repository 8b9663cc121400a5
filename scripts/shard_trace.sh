#!/bin/bash
# Kernel trace of the one-rank rehearsal of the sharded loop at the 8-way shard size (no link in it):
#   gpurun -- 'bash scripts/shard_trace.sh r03 [extra env assignments]'
# -> gpurun_out/shard_trace_<tag>/ ; the per-kernel summary goes to profiles/<tag>_sharded_one_rank_kernel_stats.csv
set -o pipefail
TAG=${1:-r03}; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/shard_trace_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export LCG_HIP_FORCE_COMM=1 LCG_HIP_DIST_MODE=2 MASTER_PORT=29561 "$@"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o prof -- \
    python3 "$REPO/bench.py" --rows 1250000 --steps 500 --warmup 5 --reps 2 --no-cpu-baseline --no-variants > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "rc $?"
python3 - "$OUT" <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/prof_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f} %')
P
