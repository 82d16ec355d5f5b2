cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tl20; mkdir -p gpurun_out/tl20
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl20 -o t -- python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic > gpurun_out/tl20/bench.json 2>/dev/null
F=$(find gpurun_out/tl20 -name "*kernel_trace.csv" | head -1)
python scripts/region_timeline.py $F 20 | tee gpurun_out/r03_region_timeline_20steps.txt
python -c "
import json; d=json.loads(open('gpurun_out/tl20/bench.json').read().strip().splitlines()[-1]); print('bench line ms_per_step', d['ms_per_step'])" | tee -a gpurun_out/r03_region_timeline_20steps.txt
find gpurun_out/tl20 -name "*.csv" -delete
