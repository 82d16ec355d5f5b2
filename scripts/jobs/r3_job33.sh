cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 2 3 4; do
  touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
  make -C gnuais_amd/csrc EXTRA=-DPLL_AHEAD_N=$k 2>&1 | grep -i "error\|warning" | head
  echo "== PLL_AHEAD $k"
  NCH=256 REPS=1 LPWS=16 timeout 200 python scripts/time_pll4.py 2>&1 | grep "^lag"
  REPS=2 LPWS=16 PVS=3,4 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
