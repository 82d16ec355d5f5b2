cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu --no-others --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['message_lines'], indent=1))"
