cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 1 0 1 0; do
  touch gnuais_amd/csrc/pll_nrzi3.hip
  make -C gnuais_amd/csrc EXTRA=-DPLL3_LONE_SCANNER=$k 2>&1 | grep -i "error" | head
  echo "== PLL3_LONE_SCANNER $k"
  REPS=3 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
touch gnuais_amd/csrc/pll_nrzi3.hip
