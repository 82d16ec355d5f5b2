cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 1; do
  touch gnuais_amd/csrc/fir_slice.hip
  make -C gnuais_amd/csrc EXTRA=-DFIR_VTAPS_12=$k 2>&1 | grep -i "error" | head
  echo "== FIR_VTAPS_12 $k"
  timeout 100 python scripts/fir_only_loop.py 1 0 100 2>&1 | tail -1
  timeout 100 python scripts/fir_only_loop.py 1 0 100 2>&1 | tail -1
  REPS=2 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
touch gnuais_amd/csrc/fir_slice.hip
