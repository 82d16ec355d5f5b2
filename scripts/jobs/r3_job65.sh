cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lag in 1 2; do
GNUAIS_K2B_LAG=$lag REPS=1 LPWS=8,16,32,64 PVS=3,51 timeout 400 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
