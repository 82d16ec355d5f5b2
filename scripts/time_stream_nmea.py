import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
for i in range(10):
    b.run(x, sync=False); b.stream_nmea(copy=False)
torch.cuda.synchronize()
t = time.perf_counter()
n = 20
for i in range(n):
    b.run(x, sync=False); b.stream_nmea(copy=False)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / n * 1e3)
