cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GNUAIS_PIPELINE=0 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -2
GNUAIS_K2B_LAG=2 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_nmea.py -m gpu -x -q 2>&1 | tail -2
GNUAIS_K2B_LAG=2 PIPE=1 timeout 200 python scripts/fuzz_parity.py 90 80000 2>&1 | tail -1
