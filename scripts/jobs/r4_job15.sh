# round 4, job 15: K2b fed segment by segment beside its own PLL launch (k2b_dataflow): parity, then timing off / on
mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/r4/job15_pytest.txt
cat gpurun_out/r4/job15_pytest.txt
rm -f gpurun_out/r4/job15_flow.txt
for f in 0 1 0 1; do
  echo "k2b_dataflow $f" >> gpurun_out/r4/job15_flow.txt
  ( GNUAIS_K2B_DATAFLOW=$f REPS=7 timeout 600 python scripts/time_sched.py 3,-1,1,1 2,-1,1,1 4,-1,1,1 2>&1 | grep -v amdgpu ) >> gpurun_out/r4/job15_flow.txt
done
cat gpurun_out/r4/job15_flow.txt
( PIPE=1 timeout 300 python scripts/fuzz_parity.py 200 470000 2>&1 | tail -2 )
( timeout 300 python scripts/fuzz_parity.py 200 480000 2>&1 | tail -2 )
