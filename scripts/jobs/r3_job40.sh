cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 2 3; do
  touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
  make -C gnuais_amd/csrc EXTRA=-DPLL_SCAN_PRIO=$k 2>&1 | grep -i "error" | head
  echo "== PLL_SCAN_PRIO $k"
  REPS=2 LPWS=16 PVS=3,32,6 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
