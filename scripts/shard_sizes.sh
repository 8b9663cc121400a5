#!/bin/bash
# One-rank rehearsal of the sharded CG loop at the shard sizes of the 2-, 4- and 8-way split of the 10M-row system (no link in it),
# next to the unsharded system on the same box:  gpurun -- 'bash scripts/shard_sizes.sh > gpurun_out/shard_sizes.txt'
set -e -o pipefail
cd ${GRAFT_REPO_ROOT:-.}
line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'K=$2', round(d['ms_per_step']*1e3,1), 'us/iteration', d.get('config',{}).get('exchange',''))"; }
for k in 20 500; do
  python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-variants --no-live-pmc 2>/dev/null | line "10000000 rows, one GPU, no communicator" $k
  for rows in 5000000 2500000 1250000; do
    LCG_HIP_FORCE_COMM=1 LCG_HIP_DIST_MODE=2 MASTER_PORT=29565 python3 bench.py --rows $rows --steps $k --warmup 5 --no-cpu-baseline --no-variants 2>/dev/null | line "$rows rows, one-rank communicator" $k
  done
done
