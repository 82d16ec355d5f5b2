cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 500 python scripts/fuzz_parity.py 420 110000 2>&1 | tail -1
PIPE=1 timeout 500 python scripts/fuzz_parity.py 420 120000 2>&1 | tail -1
DEFRAMER=1 timeout 400 python scripts/fuzz_parity.py 300 130000 2>&1 | tail -1
TABLE=192k timeout 400 python scripts/fuzz_parity.py 300 140000 2>&1 | tail -1
