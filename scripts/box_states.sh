#!/bin/bash
# The two states of a box: the headline bench six times in a row (fresh processes), clocks and power read in between.
#   gpurun -- 'bash scripts/box_states.sh > gpurun_out/box_states.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2 3 4 5 6; do
  python3 bench.py --no-cpu-baseline --no-live-pmc --no-variants --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run $i:', round(d['value'],1), 'it/s, A.x', round(d['roofline']['avg_launch_us'],1), 'us')"
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|memory)" | head -8
done
