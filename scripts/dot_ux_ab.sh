# u of the carried dot taken from the run blocks' own gathers where u == x (LCG_HIP_DOT_UX=0: read as before), headline bench, LAB build, alternating fresh processes.
#   gpurun -- 'bash scripts/dot_ux_ab.sh > gpurun_out/dot_ux_ab.txt'
cd ${GRAFT_REPO_ROOT:-.}
export LCG_HIP_LAB=1
for i in 1 2 3 4 5 6; do for z in 0 1; do
  LCG_HIP_DOT_UX=$z python3 bench.py --no-cpu-baseline --no-live-pmc --no-variants --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $i u from the gathers $z:', round(d['value'],1), 'it/s, A.x', round(d['roofline']['avg_launch_us'],1), 'us')"
done; done
