mkdir -p gpurun_out/r3
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3/trace -o t -- python $R/bench.py --no-cpu --no-others --no-e2e --steps 60 --warmup 10 > $R/gpurun_out/r3/trace_bench.json 2>/dev/null
cd $R
python scripts/pipeline_gaps.py $(ls gpurun_out/r3/trace/*kernel_trace.csv | head -1) 30 60 > gpurun_out/r3/pipeline_gaps.txt
cat gpurun_out/r3/pipeline_gaps.txt
cut -c1-400 gpurun_out/r3/trace_bench.json
rm -rf gpurun_out/r3/trace
