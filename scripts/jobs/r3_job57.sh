cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python scripts/fuzz_parity.py 120 70000 2>&1 | tail -1
GNUAIS_FIR_PERSIST=5 timeout 200 python scripts/fuzz_parity.py 90 71000 2>&1 | tail -1
TABLE=192k timeout 200 python scripts/fuzz_parity.py 60 72000 2>&1 | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
