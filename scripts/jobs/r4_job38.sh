# round 4, job 38: K3 back on its own stream while the batch is streaming (delivery loop 0.89 -> ?), GPU suite, the bench line again
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 ) > gpurun_out/r4/job38_pytest.txt
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 2>$GRAFT_REPO_ROOT/gpurun_out/r4/job38_bench.err | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r4/job38_bench.json
cd $GRAFT_REPO_ROOT
cat gpurun_out/r4/job38_pytest.txt; python -c "
import json; d=json.load(open('gpurun_out/r4/job38_bench.json')); print(d['ms_per_step'], d['steady_state']['ms_per_step'], d['end_to_end']['ms_per_step'], d['end_to_end']['with_vessel_table']['ms_per_step'], d['other_configs']['C5']['ms_per_step'], d['other_configs']['C2']['ms_per_step'])"
