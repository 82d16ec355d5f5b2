# round 4, job 43: the bench line with its stage-mask legs (the line itself says which stages set the period), smoke on the rebuilt library
mkdir -p gpurun_out/r4
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/job43_smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 2>$GRAFT_REPO_ROOT/gpurun_out/r4/job43_bench.err | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/r4/job43_bench.json
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/r4/job43_smoke.txt; tail -3 gpurun_out/r4/job43_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r4/job43_bench.json')); print(d['ms_per_step'], d['steady_state']['ms_per_step'], d['end_to_end']['ms_per_step'], d['stage_masks']); print(d['roofline']['valu_chain']['bound'], d['roofline']['valu_chain'].get('bound_why')); print(d['message_lines']['frames'], d['other_configs']['C5']['ms_per_step'])"
