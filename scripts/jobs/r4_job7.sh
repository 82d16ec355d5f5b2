# round 4, job 7: fuzz soak of the time-parallel PLL and of the round's host-side changes (carry rotation, two FIR streams)
mkdir -p gpurun_out/r4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
( PLL_VARIANT=7 timeout 400 python scripts/fuzz_parity.py 300 410000 2>&1 | tail -2 ) > gpurun_out/r4/job7_fuzz.txt
( PLL_VARIANT=7 TABLE=192k timeout 300 python scripts/fuzz_parity.py 200 420000 2>&1 | tail -2 ) >> gpurun_out/r4/job7_fuzz.txt
( PLL_VARIANT=7 PIPE=1 timeout 300 python scripts/fuzz_parity.py 200 430000 2>&1 | tail -2 ) >> gpurun_out/r4/job7_fuzz.txt
( timeout 300 python scripts/fuzz_parity.py 200 440000 2>&1 | tail -2 ) >> gpurun_out/r4/job7_fuzz.txt
( GNUAIS_FIR_STREAMS=2 GNUAIS_NBUF=6 PIPE=1 timeout 300 python scripts/fuzz_parity.py 200 450000 2>&1 | tail -2 ) >> gpurun_out/r4/job7_fuzz.txt
cat gpurun_out/r4/job7_fuzz.txt
