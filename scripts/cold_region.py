"""The driver's timed region alone, for a kernel trace: warm-up, sync, pause, N steps, sync, pause -- so that the LAST N
calls of the trace are exactly a region that started cold (scripts/region_timeline.py <trace> N).
usage: cold_region.py [steps=20]   (options through the GNUAIS_* environment variables)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(256, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
stream = torch.cuda.current_stream().cuda_stream
b = ReceiverBatch(n_ch, max_len=total)
b.autotune(x, stream)
def step():
    b.run(x, stream=stream, sync=False); b.discard_frames(stream)
for rep in range(3):
    for _ in range(5): step()
    b.sync(); torch.cuda.synchronize(); time.sleep(0.02)
    t = time.perf_counter()
    for _ in range(steps): step()
    b.sync(); torch.cuda.synchronize()
    print(f"region {rep}: {(time.perf_counter() - t) / steps * 1e3:.3f} ms/step", flush=True)
    time.sleep(0.02)
