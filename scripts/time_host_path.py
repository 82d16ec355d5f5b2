"""PCIe-inclusive rate of gnuais_batch_run_host(): pageable vs pinned source buffer."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
pageable = x.cpu().numpy()
pinned_t = torch.empty(x.shape, dtype=torch.int16).pin_memory(); pinned_t.copy_(x.cpu())
pinned = pinned_t.numpy()
b = ReceiverBatch(n_ch, max_len=total)
for name, h in (("pageable", pageable), ("pinned", pinned)):
    b.run(h); b.discard_frames()
    t = time.perf_counter()
    for _ in range(3):
        b.run(h); b.discard_frames()
    dt = (time.perf_counter() - t) / 3
    print(f"{name:9s} {dt*1e3:7.1f} ms per call  {n_ch*total/dt/1e9:6.1f} Gsamples/s  {n_ch*total*2/dt/1e9:6.1f} GB/s", flush=True)
