# round 4, job 41: fuzz soak on the final tree
mkdir -p gpurun_out/r4
f=gpurun_out/r4/job41_fuzz.txt
rm -f $f
( timeout 400 python scripts/fuzz_parity.py 300 900000 2>&1 | tail -1 ) >> $f
( PIPE=1 timeout 400 python scripts/fuzz_parity.py 240 910000 2>&1 | tail -1 ) >> $f
( DEFRAMER=1 timeout 300 python scripts/fuzz_parity.py 200 920000 2>&1 | tail -1 ) >> $f
( TABLE=192k timeout 300 python scripts/fuzz_parity.py 160 930000 2>&1 | tail -1 ) >> $f
cat $f
