cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python scripts/fuzz_parity.py 150 5000 2>&1 | tail -2
PIPE=1 timeout 300 python scripts/fuzz_parity.py 60 7000 2>&1 | tail -2
