cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python scripts/fuzz_parity.py 100 9000 2>&1 | tail -1
PIPE=1 timeout 200 python scripts/fuzz_parity.py 60 9500 2>&1 | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
