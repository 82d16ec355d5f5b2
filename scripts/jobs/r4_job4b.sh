mkdir -p gpurun_out/r4
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r4/job4_pytest.txt
bash scripts/jobs/r4_job4.sh
cat gpurun_out/r4/job4_pytest.txt
