#!/bin/bash
# Placement study, step 3: (a) every family of allocations with the two probes beside each pair (pure write of y; value stream + write
# of y) -- does the effect need the gathers? -- and (b) the LAB library (make LAB=1), the same vectors with the product's y stores
# plain / non-temporal / sc1 / sc0 sc1 -- does a store policy take the slow class away?
#   gpurun --timeout 600 -- 'bash scripts/collect_placement3.sh r04_placement3'
set -o pipefail
TAG=${1:-r04_placement3}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/$TAG
LAB=$REPO/scripts/bin/placement_lab5
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
  timeout -k 10 120 "$LAB" 4 127 10000000 1 > "$OUT/probes$i.txt" 2> "$OUT/probes$i.err"; rc=$?; echo "probes $i: rc $rc"
  if [ $rc -ne 0 ]; then echo "stopping"; exit 1; fi
done
for i in 1 2 3; do
  LD_LIBRARY_PATH=$REPO/liblcg_amd/lib/lab:$LD_LIBRARY_PATH timeout -k 10 150 "$LAB" 3 55 10000000 0 1 > "$OUT/ystore$i.txt" 2> "$OUT/ystore$i.err"; rc=$?; echo "ystore $i: rc $rc"
  if [ $rc -ne 0 ]; then echo "stopping"; exit 1; fi
done
