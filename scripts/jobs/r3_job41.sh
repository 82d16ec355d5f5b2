cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 3 0 3; do
  touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
  make -C gnuais_amd/csrc EXTRA=-DPLL_SCAN_PRIO=$k 2>&1 | grep -i "error" | head
  echo "== PLL_SCAN_PRIO $k"
  REPS=3 LPWS=16 PVS=3 timeout 300 python scripts/time_pll4.py 2>&1 | grep "^lag"
  python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 20 steps: ms_per_step %.4f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()})"
done
touch gnuais_amd/csrc/pll_nrzi.hip gnuais_amd/csrc/pll_nrzi3.hip
