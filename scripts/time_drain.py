"""How long does delivering the frames of one C3 call to the host take?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
b = ReceiverBatch(n_ch, max_len=total)
for it in range(4):
    b.run(x)
    t = time.perf_counter(); f = b.drain_frames(); dt = time.perf_counter() - t
    key = f["channel"].astype(np.int64) << 32 | f["end_bit"]
    print(f"drain: {len(f)} frames in {dt*1e3:.2f} ms, ordered={bool(np.all(np.diff(key) > 0))}")
# host-buffer path (PCIe inclusive)
xh = x[:, :2048].contiguous().cpu().numpy()
b2 = ReceiverBatch(2048, max_len=total)
for it in range(3):
    t = time.perf_counter(); b2.run(xh); dt = time.perf_counter() - t
    print(f"run_host 2048 ch x {total}: {dt*1e3:.2f} ms -> {2048*total/dt/1e6:.0f} Msamples/s (PCIe-inclusive, pageable host memory)")
