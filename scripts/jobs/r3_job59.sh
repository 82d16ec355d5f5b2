cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for T in 256 384 512 768 1024 1536; do
  echo "== GNUAIS_FIR_T $T"
  GNUAIS_FIR_T=$T REPS=2 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
done
