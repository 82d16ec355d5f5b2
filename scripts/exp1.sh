cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o tl -- python $R/scripts/exp3.py 2>&1 | grep "ms/step"
f=$(find $O -name "*kernel_trace.csv" | head -1)
python $R/scripts/timeline.py $f 70 | cut -c1-110
