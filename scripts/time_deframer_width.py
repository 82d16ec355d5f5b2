import sys,os
sys.path.insert(0,os.getcwd())
import torch, time
from gnuais_amd import ReceiverBatch, synth, tile_channels
base,_=synth.make_base_streams(64,48000,seed=synth.SEED)
x=tile_channels(torch.from_numpy(base).cuda(),16384)
for lpw in (16,8,4,2,16,8,4,2):
    b=ReceiverBatch(16384,max_len=48000); b.set_option("hdlc_lpw",lpw); b.set_timing(True)
    for i in range(6):
        b.run(x); r=b.last_timing(); b.discard_frames()
    b.set_timing(False)
    for i in range(20): b.run(x,sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(100): b.run(x,sync=False); b.discard_frames()
    b.sync(); torch.cuda.synchronize(); dt=(time.perf_counter()-t)/100
    print("lpw",lpw,"isolated",{k:round(t,3) for k,t in r.items()}, "pipelined ms/step", round(dt*1e3,4), b.total_received(), flush=True)
    del b
