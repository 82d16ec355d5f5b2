cp gnuais_amd/libgnuais_hip.so /tmp/keep.so
for v in base pad2 pad4 base; do
  if [ $v = base ]; then cp /tmp/keep.so gnuais_amd/libgnuais_hip.so; else cp scripts/ab/lib_$v.so gnuais_amd/libgnuais_hip.so; fi
  echo "== $v"; python scripts/time_fir_wide.py all 1:0:512 2>&1 | grep "cpl 1"
done
cp /tmp/keep.so gnuais_amd/libgnuais_hip.so
