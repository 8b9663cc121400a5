#!/bin/bash
# Collect, on ONE GPU box in ONE call, everything profiles/ is built from (tag = $1, default r02):
#   gpurun --timeout 1200 -- 'bash scripts/collect_profiles.sh r02'
# then, back in the build container:
#   python scripts/make_profile_summary.py r02
# Counters are collected in their own passes, never together with a trace (MI355X_MICROARCH.md); the profiled
# program stands directly behind `--`.
set -e -o pipefail
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/profiles_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO"
# 1. the bench line itself (headline + variants + cpu baseline)
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"; echo
cd /tmp && export TMPDIR=/tmp
# 2. kernel trace of the same command (the variants ride in it: one stats file covers all three patterns)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -o prof -- \
    python3 "$REPO/bench.py" --no-cpu-baseline --no-live-pmc > "$OUT/trace_bench.json" 2> "$OUT/trace_bench.err"
echo "bench kernel trace done"
# 3. kernel traces of the other single-GPU configurations: configs[1] (Laplacian PCG), configs[4](i) (10M non-symmetric
#    BiCGStab / CGS), configs[0] and the complex system (latency-bound)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_configs" -o prof -- \
    python3 "$REPO/scripts/bench_configs.py" c1 c2 c5 > "$OUT/configs_traced.jsonl" 2> "$OUT/trace_configs.err"
echo "configs kernel trace done"
# 4. counters, one pass each, on the A.x of every pattern (automatic kernel choice; --dot 1 = the product as the solver
#    loops run it, with the dot that follows it carried where the kernel family can)
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    name=${c// /_}
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$name" -o pmc -- \
        python3 "$REPO/scripts/ax_variants.py" --modes auto --reps 5 --dot 1 > "$OUT/pmc_$name.jsonl" 2> "$OUT/pmc_$name.err"
    echo "pmc $c done"
done
# 5. counters on the headline CG iteration (BLAS-1 kernels: the calibration of FETCH_SIZE on known byte counts)
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmcbench_$c" -o pmc -- \
        python3 "$REPO/bench.py" --no-cpu-baseline --no-live-pmc --no-variants --steps 20 --warmup 2 --reps 1 > /dev/null 2> "$OUT/pmcbench_$c.err"
    echo "pmc(bench) $c done"
done
cd "$REPO"
# 6. un-profiled numbers of the secondary configurations and of the A.x variants (plain / binned / tiled / automatic)
timeout -k 10 600 python3 scripts/bench_configs.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
timeout -k 10 300 python3 scripts/ax_variants.py --modes plain,binned,tiled,auto > "$OUT/ax_variants.jsonl" 2> "$OUT/ax_variants.err"
echo "configs done"
