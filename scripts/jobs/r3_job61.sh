cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
GNUAIS_PIPELINE=0 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py -m gpu -q --deselect tests/test_hip_parity.py::test_autotune_keeps_results_and_resets_state 2>&1 | tail -8
