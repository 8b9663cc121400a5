#!/bin/bash
# The two states of a box with and without the placement of the product's output (lcg_hip_set_placement): the headline
# bench in fresh processes, alternating LCG_HIP_PLACE=0 (work vectors as allocated) and the default (output roles to the vectors the
# product writes fastest).      gpurun -- 'bash scripts/box_states2.sh 5 > gpurun_out/box_states2.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
PAIRS=${1:-4}
for i in $(seq 1 $PAIRS); do
  if [ $((i % 2)) -eq 1 ]; then ORDER="0 -1"; else ORDER="-1 0"; fi
  for place in $ORDER; do
    LCG_HIP_PLACE=$place python3 bench.py --no-cpu-baseline --no-live-pmc --no-variants --steps 100 --warmup 10 2> >(grep -m2 "placement walk\|placement: [1-9]" >&2) | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d.get('placement') or {}
print('pair $i LCG_HIP_PLACE=$place:', round(d['value'],1), 'it/s, A.x', round(d['roofline']['avg_launch_us'],1), 'us; placement: timed', p.get('vectors_timed_in_first_solve'), 'moved', p.get('roles_moved'), 'output', round(p.get('first_output_us_as_allocated',0),1), '->', round(p.get('first_output_us_as_placed',0),1), 'us')"
  done
done
