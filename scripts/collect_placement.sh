#!/bin/bash
# The two states of a box, counter by counter.  scripts/bin/placement_lab5 (one process: one matrix, pairs of
# vectors allocated in five ways, every product one dispatch, a manifest that ties dispatches to pairs) first plain, then under
# `rocprofv3 --pmc` once per counter group -- separate passes, the program directly behind `--`, no trace domain beside the counters.
# Fast and slow pairs occur INSIDE each process (the first hipMalloc pair is fast, the later ones slow: profiles/r03_placement.txt),
# so every pass carries its own classification (the dispatch durations of the pass itself).
#   gpurun --timeout 1100 -- 'bash scripts/collect_placement.sh r04_placement'
# -> gpurun_out/<tag>/ ; condensed by scripts/make_placement_summary.py <tag> into profiles/<tag>_counters.csv
set -o pipefail
TAG=${1:-r04_placement}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
LAB=$REPO/scripts/bin/placement_lab5
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do      # every family of allocations, with the two probes (pure write of y; value stream + write of y) beside each pair
  timeout -k 10 120 "$LAB" 4 127 10000000 1 > "$OUT/plain$i.txt" 2> "$OUT/plain$i.err"; echo "plain $i: rc $?"
done
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum" \
         "TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
         "TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_NORMAL_WRITEBACK_sum" \
         "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
         "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "GRBM_EA_BUSY GRBM_TC_BUSY" \
         "TCC_BUSY_sum TCC_CYCLE_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
         "TCC_EA0_RDREQ TCC_EA0_WRREQ" \
         "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_IB_STALL_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o pmc -- "$LAB" 3 55 > "$OUT/pass$i.txt" 2> "$OUT/pass$i.err"
    rc=$?
    echo "pass $i ($c): rc $rc"
    echo "$c" > "$OUT/pass$i.counters"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
# keep what travels back small: the counter CSVs only
find "$OUT" -name "*.csv" ! -name "*counter_collection.csv" -delete
du -sh "$OUT"
