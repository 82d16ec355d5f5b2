"""Profiling target: a few passes of the full chain at C3 size (no CPU baseline)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch = int(os.environ.get("NCH", 16384)); total = int(os.environ.get("LEN", 48000))
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
for k, v in os.environ.items():
    if k.startswith("OPT_"):
        b.set_option(k[4:], int(v))
for it in range(int(os.environ.get("ITERS", 3))):
    b.run(x, sync=(os.environ.get("SYNC","1")=="1")); b.discard_frames()
torch.cuda.synchronize()
print("received", b.total_received())
