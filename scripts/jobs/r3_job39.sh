cd $GRAFT_REPO_ROOT
timeout 120 scripts/ubench/pll_step6.bin
