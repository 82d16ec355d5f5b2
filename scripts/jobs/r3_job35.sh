cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 1 2 3 4 7; do
  touch gnuais_amd/csrc/pll_nrzi.hip
  make -C gnuais_amd/csrc EXTRA=-DSCAN_EXP=$k 2>&1 | grep -i "error" | head
  echo "== SCAN_EXP $k"
  NCH=256 REPS=1 LPWS=16 PVS=4,6 timeout 200 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
touch gnuais_amd/csrc/pll_nrzi.hip
