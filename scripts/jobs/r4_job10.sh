mkdir -p gpurun_out/r4
( timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r4/job10_pytest.txt
cat gpurun_out/r4/job10_pytest.txt
