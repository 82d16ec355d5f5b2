cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "node" 2>&1 | tail -12
