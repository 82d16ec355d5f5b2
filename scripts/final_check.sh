#!/bin/bash
# The round's closing run on the GPU box: the GPU suite, smoke, the profiles of the bench command (scripts/collect_profiles.sh <tag>).
TAG=${1:-r06}
mkdir -p gpurun_out/$TAG
( time timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > gpurun_out/${TAG}_pytest.txt 2>&1
cat gpurun_out/${TAG}_pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
tail -1 gpurun_out/${TAG}_collect.log | cut -c1-400
