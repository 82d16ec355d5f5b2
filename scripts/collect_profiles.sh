#!/bin/bash
# Runs on the GPU box (via gpurun): bench line + rocprofv3 summaries of the SAME command.
# Counter passes are separate runs with --kernel-trace only (no --stats, no sys-trace).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py 2>$O/bench.err | tail -1 > $O/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $R/bench.py --no-cpu --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o pmc -- python $R/bench.py --no-cpu --steps 6 --warmup 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -o pmc -- python $R/bench.py --no-cpu --steps 6 --warmup 1 > /dev/null 2>&1
GNUAIS_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nopipe -o bench -- python $R/bench.py --no-cpu > $O/bench_nopipe.json 2>/dev/null
cat $O/bench.json
