cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
make -C gnuais_amd/csrc EXTRA=-DPLL_SCAN_PRIO=0 2>&1 | grep -i "error" | head
NCH=256 REPS=2 LPWS=16 PVS=3,6 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
