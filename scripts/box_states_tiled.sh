#!/bin/bash
# The tiled product's two states: the row-random band of bench.py in fresh processes, LCG_HIP_PLACE=0 against the
# default, with what the placement and its walk said.     gpurun -- 'bash scripts/box_states_tiled.sh 3 > gpurun_out/box_states_tiled.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
PAIRS=${1:-3}
for i in $(seq 1 $PAIRS); do
  if [ $((i % 2)) -eq 1 ]; then ORDER="0 -1"; else ORDER="-1 0"; fi
  for place in $ORDER; do
    LCG_HIP_DEBUG=1 LCG_HIP_PLACE=$place python3 bench.py --pattern row_random_band --no-cpu-baseline --no-live-pmc --no-variants --steps 100 --warmup 10 \
        2> >(grep "placement\|tiled plan" | head -8 >&2) | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('placement') or {}
print('pair $i LCG_HIP_PLACE=$place:', round(d['value'],1), 'it/s, A.x', round(d['roofline']['avg_launch_us'],1), 'us; placement: timed', p.get('vectors_timed_in_first_solve'), 'moved', p.get('roles_moved'), 'output', round(p.get('first_output_us_as_allocated',0),1), '->', round(p.get('first_output_us_as_placed',0),1), 'us', d['roofline']['kernel'][:40])"
  done
done
