# round 4, job 39: the stream assignment is tuned with K3 on its own stream (the delivery loop runs it there); the chain and the delivery loop again
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job39.txt
rm -f $out
( REPS=7 timeout 300 python scripts/time_sched.py 3,-1,1,1 3,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('bench: 20-step', round(d['ms_per_step'],4), 'steady', round(d['steady_state']['ms_per_step'],4), 'e2e', round(d['end_to_end']['ms_per_step'],4), 'with table', round(d['end_to_end']['with_vessel_table']['ms_per_step'],4))" >> $GRAFT_REPO_ROOT/$out
done
cd $GRAFT_REPO_ROOT
( timeout 900 python -m pytest tests/test_nmea.py tests/test_vessels.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2 ) >> $out
cat $out
