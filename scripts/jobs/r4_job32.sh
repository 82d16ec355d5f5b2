# round 4, job 32: timing experiment (wrong records): the deframer without its raw-bit copy -- what would leaving the raw bits in the
# packs (K3 reading them there) buy the pipeline?
mkdir -p gpurun_out/r4
out=gpurun_out/r4/job32.txt
rm -f $out
cp gnuais_amd/libgnuais_hip.so /tmp/lib_new.so
for rep in 1 2; do
for lib in new nocopy; do
  if [ $lib = new ]; then cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$lib.so gnuais_amd/libgnuais_hip.so; fi
  echo "lib $lib" >> $out
  ( REPS=5 timeout 300 python scripts/time_sched.py 3,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> $out
  ( timeout 300 python scripts/time_masks.py 8,4 24,4 2>&1 | grep -v amdgpu ) >> $out
done
done
cp /tmp/lib_new.so gnuais_amd/libgnuais_hip.so
cat $out
