#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 summaries of the SAME command, per round.
# Counter passes are separate runs with --kernel-trace only (no --stats, no sys-trace).
#   bash scripts/collect_profiles.sh [tag]      -> gpurun_out/<tag>/..., then scripts/summarize_profiles.py
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu --no-others --no-traffic --no-e2e"
python $R/bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
cp $R/bench_detail.json $O/bench_detail.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- $B --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- $B --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -o pmc -- $B --steps 6 --warmup 1 > /dev/null 2>&1
GNUAIS_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nopipe -o bench -- $B > $O/bench_nopipe.json 2>/dev/null
# C5 (192 kHz, 144 taps): kernel stats and traffic of its own
C5="python $R/bench.py --config C5 --no-cpu --no-traffic --no-e2e --steps 12 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5_stats -o bench -- $C5 > $O/c5_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/c5_pmc_fetch -o pmc -- $C5 --steps 4 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/c5_pmc_write -o pmc -- $C5 --steps 4 --warmup 1 > /dev/null 2>&1
# C2 (256 channels)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_stats -o bench -- python $R/bench.py --config C2 --no-cpu --no-traffic --no-e2e --steps 100 > $O/c2_bench_under_rocprof.json 2>/dev/null
find $O -name "*agent_info.csv" -delete
# the summaries are made here, next to the raw files (the per-dispatch traces are tens of MB and stay behind):
# copy gpurun_out/<tag>_summary/* into profiles/
python $R/scripts/summarize_profiles.py $O $TAG $R/gpurun_out/${TAG}_summary > /dev/null
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*domain_stats.csv" -delete
ls -R $O | head -40
cat $O/bench.json | cut -c1-600
