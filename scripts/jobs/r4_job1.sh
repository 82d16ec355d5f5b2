# round 4, job 1: the state of HEAD on today's box -- GPU suite, the driver's bench command, the 20-step region's timeline
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/job1_pytest.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/job1_bench.json 2> gpurun_out/r4/job1_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/rt -o rt --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-traffic --no-e2e --no-others --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r4/job1_rt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/rt -name '*kernel_trace.csv' | head -1)
python scripts/region_timeline.py $f 20 > gpurun_out/r4/job1_region_timeline.txt 2>&1
tail -3 gpurun_out/r4/job1_pytest.txt; cat gpurun_out/r4/job1_bench.json | cut -c1-1500; head -30 gpurun_out/r4/job1_region_timeline.txt
