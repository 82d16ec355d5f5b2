cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys,os
sys.path.insert(0,os.getcwd())
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
base,_=synth.make_base_streams(64,48000)
x=tile_channels(torch.from_numpy(base).cuda(),16384)
b=ReceiverBatch(16384,max_len=48000); b.set_timing(True)
for i in range(4):
    b.run(x); r=b.last_timing(); b.drain_frames()
print("isolated",{k:round(v,3) for k,v in r.items()})
PY
STEPS=100 SWEEP="stage_mask=31,3" python scripts/pipe_experiment.py 2>&1 | grep -v amdgpu.ids
