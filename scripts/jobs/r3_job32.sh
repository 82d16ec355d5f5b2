set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "ragged or extremes" 2>&1 | tail -5
PLL_VARIANT=4 timeout 300 python scripts/fuzz_parity.py 120 2>&1 | tail -4
timeout 400 python scripts/time_pll4.py 2>&1 | tail -14
GNUAIS_K2B_LAG=2 timeout 400 python scripts/time_pll4.py 2>&1 | tail -14
NCH=256 timeout 200 python scripts/time_pll4.py 2>&1 | tail -14
