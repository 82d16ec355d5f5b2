cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged or extremes" 2>&1 | tail -3
PLL_VARIANT=32 timeout 300 python scripts/fuzz_parity.py 60 2>&1 | tail -2
NCH=256 REPS=1 LPWS=16 PVS=3,32,6 timeout 200 python scripts/time_pll4.py 2>&1 | grep "^lag"
REPS=2 LPWS=16,32 PVS=3,32 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
GNUAIS_K2B_LAG=2 REPS=2 LPWS=16,32 PVS=3,32 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
