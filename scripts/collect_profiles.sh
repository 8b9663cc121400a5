#!/bin/bash
# Collect, on ONE GPU box in ONE call, everything profiles/ is built from (tag = $1, default r01):
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r01'
# then, back in the build container:
#   python scripts/make_profile_summary.py r01 gpurun_out/prof_r01 gpurun_out/pmc_r01_*
#   cp gpurun_out/bench_r01.json profiles/r01_bench_n1.json; cp gpurun_out/configs_r01.jsonl profiles/r01_configs.jsonl
# Counters are collected in their own passes, never together with a trace (MI355X_MICROARCH.md).
set -e -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd "$REPO"
timeout -k 10 400 python3 bench.py > "$OUT/bench_$TAG.json" 2> "$OUT/bench_$TAG.err"
tail -c 400 "$OUT/bench_$TAG.json"; echo
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_$TAG" "$OUT"/pmc_${TAG}_*
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o prof -- \
    python3 "$REPO/bench.py" --no-cpu-baseline > "$OUT/prof_${TAG}_bench.json" 2> "$OUT/prof_$TAG.err"
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    name=${c// /_}
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_${TAG}_$name" -o pmc -- \
        python3 "$REPO/bench.py" --no-cpu-baseline --steps 20 --warmup 2 > /dev/null 2> "$OUT/pmc_$name.err"
    echo "pmc $c done"
done
cd "$REPO"
timeout -k 10 600 python3 scripts/bench_configs.py > "$OUT/configs_$TAG.jsonl" 2> "$OUT/configs_$TAG.err"
echo "configs done"
