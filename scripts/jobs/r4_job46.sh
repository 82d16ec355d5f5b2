# round 4, job 46: the NMEA formatter of the delivery loop out of scratch and with the register request it uses (34 instead of 136 per wave;
# message_pack_kernel 15 instead of 104): the message-layer tests on the GPU, then the delivery loop A/B
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job46.txt
rm -f $out
( timeout 900 python -m pytest tests/test_nmea.py tests/test_vessels.py tests/test_sinks.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2 ) >> $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in before new; do
  if [ $lib = new ]; then cp /tmp/lib_new.so $GRAFT_REPO_ROOT/gnuais_amd/libgnuais_hip.so; else cp $GRAFT_REPO_ROOT/scripts/ab/lib_$lib.so $GRAFT_REPO_ROOT/gnuais_amd/libgnuais_hip.so; fi
  python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-traffic 2>/dev/null | tail -1 | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('lib $lib: 20-step', round(d['ms_per_step'],4), 'steady', round(d['steady_state']['ms_per_step'],4), 'delivered', round(d['end_to_end']['ms_per_step'],4), 'with table', round(d['end_to_end']['with_vessel_table']['ms_per_step'],4), 'lines ms', round(d['message_lines']['ms'],3))" >> $GRAFT_REPO_ROOT/$out
done
done
cp /tmp/lib_new.so $GRAFT_REPO_ROOT/gnuais_amd/libgnuais_hip.so
cat $GRAFT_REPO_ROOT/$out
