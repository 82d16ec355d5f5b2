cd $GRAFT_REPO_ROOT
run() {
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys,os
sys.path.insert(0,os.getcwd())
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
base,_=synth.make_base_streams(64,48000)
x=tile_channels(torch.from_numpy(base).cuda(),16384)
for v in (0,1):
    b=ReceiverBatch(16384,max_len=48000); b.set_option("hdlc_variant",v); b.set_timing(True)
    for i in range(4):
        b.run(x); r=b.last_timing(); b.drain_frames()
    print("variant",v,"isolated",{k:round(t,3) for k,t in r.items()}, b.total_received())
PY
STEPS=100 SWEEP="hdlc_variant=0,1;stage_mask=31" python scripts/pipe_experiment.py 2>&1 | grep -v amdgpu.ids
}
echo "== cap 72"; run
sed -i 's/__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(7, 8))) void hdlc_events_kernel(/__global__ __launch_bounds__(64) void hdlc_events_kernel(/' gnuais_amd/csrc/hdlc_events.hip
make -s -j8 -C gnuais_amd/csrc 2>&1 | grep -i "error"
echo "== no cap"; run
