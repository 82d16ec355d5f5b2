cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 3 32 4 51 52 6; do PLL_VARIANT=$v timeout 200 python scripts/fuzz_parity.py 100 $((90000 + v * 1000)) 2>&1 | tail -1; done
GNUAIS_FIR_CPL=2 GNUAIS_FIR_FORM=1 timeout 200 python scripts/fuzz_parity.py 100 97000 2>&1 | tail -1
GNUAIS_FIR_PK=1 timeout 200 python scripts/fuzz_parity.py 100 98000 2>&1 | tail -1
