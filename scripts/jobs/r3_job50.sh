cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NCH=256 REPS=2 LPWS=16 PVS=3,32 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
REPS=3 LPWS=16 PVS=3 timeout 400 python scripts/time_pll4.py 2>&1 | grep "^lag"
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
PLL_VARIANT=3 timeout 200 python scripts/fuzz_parity.py 90 60000 2>&1 | tail -1
