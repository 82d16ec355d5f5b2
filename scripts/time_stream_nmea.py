"""End-to-end loop (run + stream_nmea every step) with the host time spent inside each of the two calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
if os.environ.get("AUTOTUNE", "1") == "1":
    b.autotune(x)
if os.environ.get("AUTOTUNE_DELIVERY", "1") == "1":
    print("autotune_delivery: best ms per call", b.autotune_delivery(x))
for i in range(20):
    b.run(x, sync=False); b.stream_nmea(copy=False)
torch.cuda.synchronize()
n = int(os.environ.get("STEPS", 100))
for rep in range(3):
    tr = ts = 0.0
    t = time.perf_counter()
    for i in range(n):
        a = time.perf_counter(); b.run(x, sync=False); c = time.perf_counter(); b.stream_nmea(copy=False); d = time.perf_counter()
        tr += c - a; ts += d - c
    torch.cuda.synchronize()
    print("ms/step", (time.perf_counter() - t) / n * 1e3, "host in run()", tr / n * 1e3, "host in stream_nmea()", ts / n * 1e3, flush=True)
# the same loop without delivery, for comparison
for i in range(20):
    b.run(x, sync=False); b.discard_frames()
torch.cuda.synchronize()
tr = 0.0
t = time.perf_counter()
for i in range(n):
    a = time.perf_counter(); b.run(x, sync=False); b.discard_frames(); tr += time.perf_counter() - a
torch.cuda.synchronize()
print("kernel-only ms/step", (time.perf_counter() - t) / n * 1e3, "host", tr / n * 1e3)
