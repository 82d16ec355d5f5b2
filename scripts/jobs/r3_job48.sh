cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
REPS=3 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
GNUAIS_K2B_LAG=2 REPS=3 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
GNUAIS_K2B_LAG=2 REPS=2 LPWS=32 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
