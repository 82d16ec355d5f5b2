cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GNUAIS_FIR_PERSIST=5 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -x -q -k "not node and not bench and not shim and not dropin" 2>&1 | tail -2
for pz in 0 5 4 6; do
  echo "== GNUAIS_FIR_PERSIST $pz"
  for r in 1 2; do GNUAIS_FIR_PERSIST=$pz python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-e2e --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('20 steps: %.4f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernel_ms'].items()}, 'steady', d['steady_state']['ms_per_step'] if d.get('steady_state') else None)"; done
done
