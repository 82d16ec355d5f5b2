cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-traffic 2>gpurun_out/bench44.err | tail -1 > gpurun_out/bench44.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench44.json').read())
print('ms_per_step', d['ms_per_step'])
print(json.dumps(d.get('end_to_end'), indent=1))
PY
tail -3 gpurun_out/bench44.err
