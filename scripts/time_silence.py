"""K1s on digital silence vs on the synthetic workload (isolated kernel times, ms)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnuais_amd import ReceiverBatch, synth, tile_channels
n_ch, total = 16384, 48000
base, _ = synth.make_base_streams(64, total)
live = tile_channels(torch.from_numpy(base).cuda(), n_ch)
half = live.clone(); half[:, ::2] = 0          # every other channel silent
inputs = {"live": live, "silent": torch.zeros_like(live), "half silent": half}
for name, x in inputs.items():
    b = ReceiverBatch(n_ch, max_len=total)
    b.set_timing(True)
    ts = []
    for _ in range(4):
        b.run(x, sync=True); b.discard_frames()
        ts.append(b.last_timing()["fir_slice"])
    print(f"{name:12s} fir_slice ms {min(ts):.3f}", flush=True)
    del b
