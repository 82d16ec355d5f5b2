"""FIR alone, back to back, for inputs that do / do not fit the 256 MB memory-side cache."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
for n_ch, total in ((16384, 48000), (16384, 12000), (16384, 6000), (16384, 3000), (4096, 48000), (4096, 12000)):
    base, _ = synth.make_base_streams(64, total)
    x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_option("stage_mask", 1)
    for _ in range(5):
        b.run(x, sync=False)
    b.sync(); torch.cuda.synchronize()
    steps = 60
    t = time.perf_counter()
    for _ in range(steps):
        b.run(x, sync=False)
    b.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(f"{n_ch} x {total}: {n_ch*total*2/1e6:7.1f} MB  {dt*1e3:.3f} ms/step  {n_ch*total*2/dt/1e12:.2f} TB/s", flush=True)
    del b, x
