python scripts/debug_pk.py 2>&1 | grep -v amdgpu
