#!/bin/bash
# SQ-side counters of one A.x kernel family on one column pattern (what the waves wait on): separate --pmc passes, the
# profiled program directly behind `--`, no trace domain beside the counters.
#   gpurun --timeout 900 -- 'bash scripts/collect_sq.sh r03_tiled_sq 2 tiled'
# -> gpurun_out/<tag>/pass*/... ; condensed by scripts/make_sq_summary.py <tag> into profiles/<tag>.csv
set -o pipefail
TAG=${1:-r03_tiled_sq}
PAT=${2:-2}
MODE=${3:-tiled}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o pmc -- \
        python3 "$REPO/scripts/ax_variants.py" --patterns "$PAT" --modes "$MODE" --reps 5 > "$OUT/pass$i.jsonl" 2> "$OUT/pass$i.err"
    rc=$?
    echo "pass $i ($c): rc $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
cd "$REPO"
timeout -k 10 300 python3 scripts/ax_variants.py --modes plain,binned,tiled,auto > "$OUT/ax_variants.jsonl" 2> "$OUT/ax_variants.err"
echo "ax_variants rc $?"
