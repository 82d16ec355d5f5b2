cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys,os
sys.path.insert(0,os.getcwd())
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
for nb in (64, 256):
    base,_=synth.make_base_streams(nb,48000,seed=synth.SEED)
    x=tile_channels(torch.from_numpy(base).cuda(),16384)
    for v in (0,1):
        b=ReceiverBatch(16384,max_len=48000); b.set_option("hdlc_variant",v); b.set_timing(True)
        for i in range(6):
            b.run(x); r=b.last_timing(); b.discard_frames()
        print("bases",nb,"variant",v,"isolated",{k:round(t,3) for k,t in r.items()}, b.total_received())
PY
python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -2
DEFRAMER=1 timeout 100 python scripts/fuzz_parity.py 60 27000 2>&1 | tail -1
