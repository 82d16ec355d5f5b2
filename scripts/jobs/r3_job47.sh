cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 400 python scripts/fuzz_parity.py 300 20000 2>&1 | tail -1
TABLE=192k timeout 300 python scripts/fuzz_parity.py 200 30000 2>&1 | tail -1
PIPE=1 timeout 300 python scripts/fuzz_parity.py 200 40000 2>&1 | tail -1
DEFRAMER=1 timeout 200 python scripts/fuzz_parity.py 120 50000 2>&1 | tail -1
