timeout 1500 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "shim or level_log or dropin" 2>&1 | tail -15
