# round 5, job 15: the GPU suite again (after the test fixes), fuzz, then job 14's experiments
mkdir -p gpurun_out/r5
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r5/job15_pytest.txt 2>&1
cat gpurun_out/r5/job15_pytest.txt
( timeout 200 python scripts/fuzz_parity.py 45 9000 2>&1 | tail -2 ) > gpurun_out/r5/job15_fuzz.txt
( DEFRAMER=1 timeout 200 python scripts/fuzz_parity.py 20 9700 2>&1 | tail -2 ) >> gpurun_out/r5/job15_fuzz.txt
( TABLE=192k timeout 200 python scripts/fuzz_parity.py 20 9800 2>&1 | tail -2 ) >> gpurun_out/r5/job15_fuzz.txt
cat gpurun_out/r5/job15_fuzz.txt
bash scripts/jobs5/job14.sh
