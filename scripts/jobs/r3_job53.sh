cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SECONDS=0; python bench.py 2>gpurun_out/bench53.err | tail -1 > gpurun_out/bench53.json
echo "elapsed $SECONDS s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench53.json').read())
print(d['ms_per_step'], d['value'], d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['chain']['frac'],3), d['roofline']['traffic'])
print({k:round(v['ms_per_step'],3) for k,v in d['other_configs'].items()})
print(d['steady_state']['ms_per_step'], d['end_to_end']['ms_per_step'], d['end_to_end']['with_vessel_table']['ms_per_step'])
PY
