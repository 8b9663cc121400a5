#!/bin/bash
# The product of an all-zero initial guess, made (the reference's lcg.cpp:168) or not (solvers_real.hip: ax_setup): the driver's own command
# (K = 20, W = 5) in fresh processes, alternating, LAB build (LCG_HIP_ZERO_GUESS=0 makes the product as before).
#   gpurun -- 'bash scripts/zero_guess_ab.sh 3 > gpurun_out/zero_guess_ab.txt 2>&1'
cd ${GRAFT_REPO_ROOT:-.}
export LCG_HIP_LAB=1
for i in $(seq 1 ${1:-3}); do
  for z in 0 1; do
    for rows in 10000000 1250000; do
      if [ $rows -eq 1250000 ]; then export LCG_HIP_FORCE_COMM=1 LCG_HIP_DIST_MODE=2; else unset LCG_HIP_FORCE_COMM LCG_HIP_DIST_MODE; fi
      LCG_HIP_ZERO_GUESS=$z python3 bench.py --rows $rows --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pair $i rows $rows zero-guess shortcut $z:', round(d['value'],1), 'it/s', round(1e3*d['ms_per_step'],1), 'us/iteration')"
    done
  done
done
