cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NCH=256 REPS=1 LPWS=16 PVS=3,4,51,52,6 timeout 200 python scripts/time_pll4.py 2>&1 | grep "^lag"
REPS=2 LPWS=16 PVS=3,4,51,52,6 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
